"""TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.

Restatement of the reference's multi-scale test pipeline, /root/reference/tool/test.py:
  net_process   :122-146   (normalise, [x, flip(x)], model, softmax, flip average)
  scale_process :149-178   (mean pad, sliding crops, float64 canvases, /count, un-pad, resize back)
  test          :191-204   (image pyramid, sum over scales, argmax)
cv2 is not installed in this image: `cv2.resize(float32, INTER_LINEAR)` (test.py:177,201) is restated as
torch's half-pixel bilinear without antialiasing, which is the same formula (fx = (dx+.5)*scale-.5, edge
clamp); `cv2.copyMakeBorder(..., BORDER_CONSTANT, value=mean)` as an explicit fill.  The model is any
callable NCHW -> logits (tests pass oracle/segnet.forward).  Parity of this file is pinned only by that
restatement ("parity unpinned" for the two cv2 calls).
"""
import numpy as np
import torch
import torch.nn.functional as F


def cv2_resize_linear(img_hwc, new_w, new_h):
    t = torch.from_numpy(np.ascontiguousarray(img_hwc)).permute(2, 0, 1)[None]
    dt = t.dtype
    out = F.interpolate(t.double(), size=(new_h, new_w), mode="bilinear", align_corners=False)
    return out[0].permute(1, 2, 0).to(dt).numpy()


def net_process(model, image, mean, std=None, flip=True):
    inp = torch.from_numpy(image.transpose((2, 0, 1))).float()
    if std is None:
        for t, m in zip(inp, mean):
            t.sub_(m)
    else:
        for t, m, s in zip(inp, mean, std):
            t.sub_(m).div_(s)
    inp = inp.unsqueeze(0)
    if flip:
        inp = torch.cat([inp, inp.flip(3)], 0)
    with torch.no_grad():
        out = model(inp)
    if out.shape[2:] != inp.shape[2:]:
        out = F.interpolate(out, inp.shape[2:], mode="bilinear", align_corners=True)
    out = F.softmax(out, dim=1)
    out = (out[0] + out[1].flip(2)) / 2 if flip else out[0]
    return out.numpy().transpose(1, 2, 0)


def scale_process(model, image, classes, crop_h, crop_w, h, w, mean, std=None, stride_rate=2 / 3):
    ori_h, ori_w, _ = image.shape
    pad_h, pad_w = max(crop_h - ori_h, 0), max(crop_w - ori_w, 0)
    ph, pw = int(pad_h / 2), int(pad_w / 2)
    if pad_h > 0 or pad_w > 0:
        padded = np.empty((ori_h + pad_h, ori_w + pad_w, 3), dtype=image.dtype)
        padded[...] = np.asarray(mean, dtype=image.dtype)
        padded[ph:ph + ori_h, pw:pw + ori_w] = image
        image = padded
    new_h, new_w, _ = image.shape
    stride_h, stride_w = int(np.ceil(crop_h * stride_rate)), int(np.ceil(crop_w * stride_rate))
    grid_h = int(np.ceil(float(new_h - crop_h) / stride_h) + 1)
    grid_w = int(np.ceil(float(new_w - crop_w) / stride_w) + 1)
    pred = np.zeros((new_h, new_w, classes), dtype=float)
    cnt = np.zeros((new_h, new_w), dtype=float)
    for ih in range(grid_h):
        for iw in range(grid_w):
            s_h = ih * stride_h
            e_h = min(s_h + crop_h, new_h)
            s_h = e_h - crop_h
            s_w = iw * stride_w
            e_w = min(s_w + crop_w, new_w)
            s_w = e_w - crop_w
            crop = image[s_h:e_h, s_w:e_w].copy()
            cnt[s_h:e_h, s_w:e_w] += 1
            pred[s_h:e_h, s_w:e_w, :] += net_process(model, crop, mean, std)
    pred /= np.expand_dims(cnt, 2)
    pred = pred[ph:ph + ori_h, pw:pw + ori_w]
    return cv2_resize_linear(pred, w, h)


def multi_scale_predict(model, image, classes, base_size, crop_h, crop_w, scales, mean, std):
    """image: float32 [H,W,3].  Returns (argmax [H,W], prob [H,W,classes] float64)."""
    h, w, _ = image.shape
    prediction = np.zeros((h, w, classes), dtype=float)
    for scale in scales:
        long_size = round(scale * base_size)
        new_h = new_w = long_size
        if h > w:
            new_w = round(long_size / float(h) * w)
        else:
            new_h = round(long_size / float(w) * h)
        image_scale = cv2_resize_linear(image, new_w, new_h)
        prediction += scale_process(model, image_scale, classes, crop_h, crop_w, h, w, mean, std)
    prediction /= len(scales)
    return np.argmax(prediction, axis=2), prediction
