"""TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.

numpy restatement of the OpenCV calls the reference's input pipeline makes (/root/reference/util/transform.py:71-72,
101-102, 155-156, 192-194, 204-205, 215-216, 226, 233, 240; util/dataset.py:63-66).  OpenCV is a third-party
dependency that is NOT vendored under /root/reference and NOT installed in this image (`import cv2` fails; the
reference pins no version — README "tested with pytorch 1.4.0", i.e. opencv-python 4.1/4.2 era).  What is restated is
the published algorithm of OpenCV 4.x `modules/imgproc` for exactly the argument combinations the reference uses:

  resize(float32 HxWx3, INTER_LINEAR)      resize.cpp  resizeGeneric_ / HResizeLinear / VResizeLinear (float path)
  resize(uint8 HxW, INTER_NEAREST)         resize.cpp  resizeNN
  getRotationMatrix2D                      imgwarp.cpp
  warpAffine(INTER_LINEAR|INTER_NEAREST, BORDER_CONSTANT)
                                           imgwarp.cpp WarpAffineInvoker (AB_BITS=10 fixed-point coordinates,
                                           INTER_BITS=5 sub-pixel table) + remapBilinear / remapNearest
  GaussianBlur(float32, (k,k), 0)          smooth.dispatch.cpp: small_gaussian_tab for k<=7, separable symmetric
                                           row/column filter, BORDER_REFLECT_101
  copyMakeBorder(BORDER_CONSTANT), flip, cvtColor(RGB2BGR/BGR2RGB)

**parity unpinned** for these primitives: no cv2 binary is available to run them against.
Cross-checks that ARE made (tests/test_transform_cpu.py): resize-linear against torch's half-pixel bilinear,
blur against scipy.ndimage.correlate1d(mode="mirror"), rotate against scipy.ndimage.affine_transform (loose: the
fixed-point grid is 1/32 px).  Builds that route these calls through IPP or FMA-contracted SIMD differ from this
restatement in the last bit of the float results; integer/label results do not depend on that.
"""
import math

import numpy as np

INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT_101 = 0, 4
COLOR_BGR2RGB, COLOR_RGB2BGR = 4, 4
IMREAD_COLOR, IMREAD_GRAYSCALE = 1, 0

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, INTER_TAB_SIZE = 1 << AB_BITS, 1 << INTER_BITS

SMALL_GAUSSIAN_TAB = {
    1: [1.0],
    3: [0.25, 0.5, 0.25],
    5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
    7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125],
}


def cv_round(v):
    """saturate_cast<int>(double) == cvRound: round half to even."""
    return np.rint(v).astype(np.int64)


def resize_dsize(w, h, fx, fy):
    """resize.cpp: dsize = Size(saturate_cast<int>(ssize.width*inv_scale_x), ...) when dsize is empty."""
    return int(cv_round(w * float(fx))), int(cv_round(h * float(fy)))


def _linear_coeffs(n_dst, n_src, scale):
    """resize.cpp (INTER_LINEAR branch): fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx; clamp."""
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= n_src - 1
    f[hi], s[hi] = 0.0, n_src - 1
    s1 = np.minimum(s + 1, n_src - 1)       # weight is 0 there; the reference reads only S[sx] for dx >= xmax
    return s, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize(src, dsize, fx=None, fy=None, interpolation=INTER_LINEAR):
    h, w = src.shape[:2]
    if dsize is None:
        inv_x, inv_y = float(fx), float(fy)
        dw, dh = resize_dsize(w, h, fx, fy)
    else:
        dw, dh = int(dsize[0]), int(dsize[1])
        inv_x, inv_y = dw / w, dh / h
    assert dw > 0 and dh > 0
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if interpolation == INTER_NEAREST:
        # resizeNN: sx = min(cvFloor(x*ifx), ssize.width-1)
        xs = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * scale_x).astype(np.int64), w - 1)
        ys = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * scale_y).astype(np.int64), h - 1)
        return np.ascontiguousarray(src[ys][:, xs])
    assert interpolation == INTER_LINEAR and src.dtype == np.float32
    x0, x1, a0, a1 = _linear_coeffs(dw, w, scale_x)
    y0, y1, b0, b1 = _linear_coeffs(dh, h, scale_y)
    a0, a1 = (a0[None, :, None], a1[None, :, None]) if src.ndim == 3 else (a0[None, :], a1[None, :])
    rows = (src[:, x0] * a0 + src[:, x1] * a1).astype(np.float32)           # HResizeLinear, float accumulators
    b0, b1 = (b0[:, None, None], b1[:, None, None]) if src.ndim == 3 else (b0[:, None], b1[:, None])
    return (rows[y0] * b0 + rows[y1] * b1).astype(np.float32)               # VResizeLinear


def getRotationMatrix2D(center, angle, scale):
    """imgwarp.cpp: center is a Point2f; everything else double."""
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    ang = angle * math.pi / 180.0
    alpha, beta = math.cos(ang) * scale, math.sin(ang) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def invert_affine(M):
    """warpAffine without WARP_INVERSE_MAP (imgwarp.cpp): in-place 2x3 inversion in double."""
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(-1)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def affine_fixed_coords(m, dw, dh, interpolation):
    """WarpAffineInvoker: integer source coordinates in units of 1/1024 px, already shifted to the
    interpolation grid.  Returns X, Y int64 [dh, dw] (NEAREST: pixel index; LINEAR: 1/32 px units)."""
    x = np.arange(dw, dtype=np.float64)
    y = np.arange(dh, dtype=np.float64)
    adelta = cv_round(m[0] * x * AB_SCALE)
    bdelta = cv_round(m[3] * x * AB_SCALE)
    round_delta = AB_SCALE // 2 if interpolation == INTER_NEAREST else AB_SCALE // INTER_TAB_SIZE // 2
    X0 = cv_round((m[1] * y + m[2]) * AB_SCALE) + round_delta
    Y0 = cv_round((m[4] * y + m[5]) * AB_SCALE) + round_delta
    shift = AB_BITS if interpolation == INTER_NEAREST else AB_BITS - INTER_BITS
    X = (X0[:, None] + adelta[None, :]) >> shift
    Y = (Y0[:, None] + bdelta[None, :]) >> shift
    return X, Y


def _sat_short(v):
    return np.clip(v, -32768, 32767)


def warpAffine(src, M, dsize, flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0):
    assert borderMode == BORDER_CONSTANT
    dw, dh = int(dsize[0]), int(dsize[1])
    h, w = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    bv = np.zeros(4, dtype=np.float64)
    vals = np.atleast_1d(np.asarray(borderValue, dtype=np.float64))
    bv[:min(4, len(vals))] = vals[:4]
    if src.dtype == np.uint8:
        cval = np.clip(np.rint(bv[:cn]), 0, 255).astype(np.uint8)
    else:
        cval = bv[:cn].astype(src.dtype)
    m = invert_affine(M)
    X, Y = affine_fixed_coords(m, dw, dh, flags)
    s3 = src.reshape(h, w, cn)
    if flags == INTER_NEAREST:
        sx, sy = _sat_short(X), _sat_short(Y)
        inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        out = np.empty((dh, dw, cn), dtype=src.dtype)
        out[...] = cval
        out[inside] = s3[sy[inside], sx[inside]]
        return out.reshape(dh, dw) if src.ndim == 2 else out
    assert flags == INTER_LINEAR and src.dtype == np.float32
    sx, sy = _sat_short(X >> INTER_BITS), _sat_short(Y >> INTER_BITS)
    fx = ((X & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE)).astype(np.float32)
    fy = ((Y & (INTER_TAB_SIZE - 1)).astype(np.float32) * np.float32(1.0 / INTER_TAB_SIZE)).astype(np.float32)
    one = np.float32(1.0)
    # initInterTab2D: w[k1][k2] = vtab[k1] * htab[k2] in float
    w00, w01 = ((one - fy) * (one - fx)).astype(np.float32), ((one - fy) * fx).astype(np.float32)
    w10, w11 = (fy * (one - fx)).astype(np.float32), (fy * fx).astype(np.float32)

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        v = np.empty((dh, dw, cn), dtype=np.float32)
        v[...] = cval
        v[ok] = s3[yy[ok], xx[ok]]
        return v

    v0, v1, v2, v3 = tap(sy, sx), tap(sy, sx + 1), tap(sy + 1, sx), tap(sy + 1, sx + 1)
    acc = (v0 * w00[..., None]).astype(np.float32)
    acc = (acc + (v1 * w01[..., None]).astype(np.float32)).astype(np.float32)
    acc = (acc + (v2 * w10[..., None]).astype(np.float32)).astype(np.float32)
    acc = (acc + (v3 * w11[..., None]).astype(np.float32)).astype(np.float32)
    # remapBilinear: a pixel whose 2x2 footprint is entirely outside takes the border value itself
    gone = (sx >= w) | (sx + 1 < 0) | (sy >= h) | (sy + 1 < 0)
    acc[gone] = cval
    return acc.reshape(dh, dw) if src.ndim == 2 else acc


def border_reflect_101(p, n):
    """borderInterpolate(p, len, BORDER_REFLECT_101)."""
    p = np.asarray(p, dtype=np.int64).copy()
    if n == 1:
        return np.zeros_like(p)
    while True:
        neg, big = p < 0, p >= n
        if not (neg.any() or big.any()):
            return p
        p[neg] = -p[neg]
        p[big] = 2 * n - 2 - p[big]


def _symm_filter(a, kern, axis):
    """SymmRowSmallFilter / SymmColumnFilter, float: s = c*k0 + (p1a+p1b)*k1 + (p2a+p2b)*k2 ..., REFLECT_101."""
    n = a.shape[axis]
    half = len(kern) // 2
    k = [np.float32(v) for v in kern]
    idx = np.arange(n)

    def take(off):
        return np.take(a, border_reflect_101(idx + off, n), axis=axis)

    acc = (take(0) * k[half]).astype(np.float32)
    for j in range(1, half + 1):
        pair = (take(-j) + take(j)).astype(np.float32)
        acc = (acc + (pair * k[half + j]).astype(np.float32)).astype(np.float32)
    return acc


def GaussianBlur(src, ksize, sigmaX, sigmaY=0, borderType=BORDER_REFLECT_101):
    assert src.dtype == np.float32 and sigmaX == 0 and sigmaY == 0 and borderType == BORDER_REFLECT_101
    kw, kh = int(ksize[0]), int(ksize[1])
    if kw not in SMALL_GAUSSIAN_TAB or kh not in SMALL_GAUSSIAN_TAB:
        raise NotImplementedError("sigma=0 kernels are tabulated for ksize 1,3,5,7 only")
    if kw == 1 and kh == 1:
        return src.copy()
    rows = _symm_filter(src, SMALL_GAUSSIAN_TAB[kw], axis=1)
    return _symm_filter(rows, SMALL_GAUSSIAN_TAB[kh], axis=0)


def flip(src, flipCode):
    if flipCode == 1:
        return np.ascontiguousarray(src[:, ::-1])
    if flipCode == 0:
        return np.ascontiguousarray(src[::-1])
    return np.ascontiguousarray(src[::-1, ::-1])


def copyMakeBorder(src, top, bottom, left, right, borderType, value=0):
    assert borderType == BORDER_CONSTANT
    h, w = src.shape[:2]
    out = np.empty((h + top + bottom, w + left + right) + src.shape[2:], dtype=src.dtype)
    if src.ndim == 3:
        v = np.zeros(4, dtype=np.float64)
        vals = np.atleast_1d(np.asarray(value, dtype=np.float64))
        v[:min(4, len(vals))] = vals[:4]
        out[...] = v[:src.shape[2]].astype(src.dtype)
    else:
        v = float(np.atleast_1d(np.asarray(value, dtype=np.float64))[0])
        out[...] = np.clip(np.rint(v), 0, 255).astype(np.uint8) if src.dtype == np.uint8 else src.dtype.type(v)
    out[top:top + h, left:left + w] = src
    return out


def cvtColor(src, code):
    assert code == COLOR_BGR2RGB and src.ndim == 3 and src.shape[2] == 3
    return np.ascontiguousarray(src[:, :, ::-1])
