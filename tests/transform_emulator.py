"""TEST INFRASTRUCTURE: executes the stage schedule of semseg_amd.transform.Compose.schedule() on the CPU with the
oracle's cv2 restatements, region by region, to check the HOST logic (stage grouping, index-map composition, needed-
region propagation) without a GPU.  Every stage sees its input only through the region its producer materialised:
the rest of the virtual image is NaN (image) / -1 (label), so a region that is too small shows up as a poisoned
output instead of going unnoticed."""
import numpy as np

from oracle import cv2_restated as ocv


def _embed(region, roi, H, W):
    """(img [h,w,3] float32, lab [h,w] int16) of a region -> poisoned full-size canvases"""
    img, lab = region
    y0, x0, h, w = roi
    ci = np.full((H, W, 3), np.nan, dtype=np.float32)
    cl = np.full((H, W), -1, dtype=np.int16)
    if h > 0 and w > 0:
        ci[y0:y0 + h, x0:x0 + w] = img
        cl[y0:y0 + h, x0:x0 + w] = lab
    return ci, cl


def _nearest_i16(lab, fn):
    """run a uint8 cv2 label op on an int16 canvas that may hold -1 (poison): shift into uint16-safe range"""
    return fn(lab)


def run_stage(stg, region, roi_in):
    k = stg["k"]
    H, W = stg["in_h"], stg["in_w"]
    ci, cl = _embed(region, roi_in, H, W)
    if k == "resize":
        oh, ow = stg["out_h"], stg["out_w"]
        x0, x1, a0, a1 = ocv._linear_coeffs(ow, W, stg["scale_x"])
        y0, y1, b0, b1 = ocv._linear_coeffs(oh, H, stg["scale_y"])
        # weights of exactly 0 must not propagate poison: cv2 reads S[sx] only there (see _linear_coeffs)
        left, right = ci[:, x0], ci[:, x1]
        rows = np.where(a1[None, :, None] == 0, left * a0[None, :, None],
                        (left * a0[None, :, None] + right * a1[None, :, None]).astype(np.float32)).astype(np.float32)
        top, bot = rows[y0], rows[y1]
        oi = np.where(b1[:, None, None] == 0, top * b0[:, None, None],
                      (top * b0[:, None, None] + bot * b1[:, None, None]).astype(np.float32)).astype(np.float32)
        xs = np.minimum(np.floor(np.arange(ow, dtype=np.float64) * stg["scale_x"]).astype(np.int64), W - 1)
        ys = np.minimum(np.floor(np.arange(oh, dtype=np.float64) * stg["scale_y"]).astype(np.int64), H - 1)
        ol = cl[ys][:, xs]
    elif k == "rotate":
        m = stg["m"]
        X, Y = ocv.affine_fixed_coords(m, W, H, ocv.INTER_NEAREST)
        sx, sy = np.clip(X, -32768, 32767), np.clip(Y, -32768, 32767)
        inside = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        ol = np.full((H, W), int(stg["pad_lab"]), dtype=np.int16)
        ol[inside] = cl[sy[inside], sx[inside]]
        X, Y = ocv.affine_fixed_coords(m, W, H, ocv.INTER_LINEAR)
        sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
        fx = ((X & 31).astype(np.float32) * np.float32(1 / 32)).astype(np.float32)
        fy = ((Y & 31).astype(np.float32) * np.float32(1 / 32)).astype(np.float32)
        one = np.float32(1)
        ws = [((one - fy) * (one - fx)), ((one - fy) * fx), (fy * (one - fx)), (fy * fx)]
        cval = np.asarray(stg["pad"], dtype=np.float32)
        acc = None
        for (dy, dx), wgt in zip(((0, 0), (0, 1), (1, 0), (1, 1)), ws):
            yy, xx = sy + dy, sx + dx
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = np.empty((H, W, 3), dtype=np.float32)
            v[...] = cval
            v[ok] = ci[yy[ok], xx[ok]]
            term = (v * wgt.astype(np.float32)[..., None]).astype(np.float32)
            acc = term if acc is None else (acc + term).astype(np.float32)
        gone = (sx >= W) | (sx + 1 < 0) | (sy >= H) | (sy + 1 < 0)
        acc[gone] = cval
        oi = acc
    elif k == "blur":
        kk = stg["ksize"]
        oi = ocv.GaussianBlur(ci, (kk, kk), 0)
        ol = cl
    else:  # gather
        oh, ow = stg["out_h"], stg["out_w"]
        yy, xx = np.mgrid[0:oh, 0:ow]
        oi = np.empty((oh, ow, 3), dtype=np.float32)
        ol = np.empty((oh, ow), dtype=np.int16)
        filled = np.zeros((oh, ow), dtype=bool)
        swapped = False
        for mp in reversed(stg["maps"]):
            swapped ^= bool(mp["swap"])
            yy, xx = mp["sy"] * yy + mp["oy"], mp["sx"] * xx + mp["ox"]
            out = ((yy < 0) | (yy >= mp["in_h"]) | (xx < 0) | (xx >= mp["in_w"])) & ~filled
            pad = np.asarray(mp["pad"], dtype=np.float32)
            oi[out] = pad[::-1] if swapped else pad
            ol[out] = min(max(int(mp["pad_lab"]), 0), 255)
            filled |= out
            yy, xx = np.where(filled, 0, yy), np.where(filled, 0, xx)
        src = ci[yy, xx]
        oi[~filled] = (src[..., ::-1] if swapped else src)[~filled]
        ol[~filled] = cl[yy, xx][~filled]
    y0, x0, h, w = stg["dst_roi"]
    return (oi[y0:y0 + h, x0:x0 + w].copy(), ol[y0:y0 + h, x0:x0 + w].copy())


def run_chain(plan, chain, image_u8, label_u8):
    """-> what the device path returns for this sample (float CHW + int64, or float HWC + uint8)"""
    H, W = label_u8.shape
    region, roi = (np.float32(image_u8), label_u8.astype(np.int16)), (0, 0, H, W)
    for stg in chain:
        ny0, nx0, nh, nw = stg["src_need"]
        y0, x0, h, w = roi
        assert nh == 0 or (ny0 >= y0 and nx0 >= x0 and ny0 + nh <= y0 + h and nx0 + nw <= x0 + w)
        region = run_stage(stg, region, roi)
        roi = stg["dst_roi"]
    img, lab = region
    assert not np.isnan(img).any(), "a stage read outside the region its producer materialised (image)"
    assert (lab >= 0).all(), "a stage read outside the region its producer materialised (label)"
    if plan.tensor:
        out = np.ascontiguousarray(img.transpose(2, 0, 1))
        if plan.norm is not None:
            mean, std = plan.norm
            for c in range(3):
                out[c] = out[c] - np.float32(mean[c])
                if std is not None:
                    out[c] = out[c] / np.float32(std[c])
        return out, lab.astype(np.int64)
    return img, lab.astype(np.uint8)
