"""Segmentation transforms with the reference's API (util/transform.py: Compose, ToTensor, Normalize, Resize,
RandScale, Crop, RandRotate, RandomHorizontalFlip, RandomVerticalFlip, RandomGaussianBlur, RGB2BGR, BGR2RGB),
executed on the device by semseg_amd/csrc/augment.hip instead of cv2 on CPU workers.

How it differs from the reference underneath the same class names:
  * a transform does not touch pixels when called; it appends to a per-sample PLAN.  Every random parameter the
    reference draws (`random.random()` / `random.randint`, in the same order, so seeding `random` reproduces the
    reference's parameters) depends only on image sizes, so the whole chain is planned before any pixel moves;
  * the final crop window is propagated backwards through flip / blur / rotate / scale and each stage computes only
    the region that can reach the output;
  * the decoded image stays uint8 until the first arithmetic stage; flip, pad, crop, channel order, ToTensor and
    Normalize are fused into the last gather, which writes float [3,h,w] / int64 [h,w] (a slot of the batch);
  * `Compose.batch(images, labels)` runs a whole batch with one launch per stage round.

`Compose(...)(image, label)` keeps the reference call shape for one sample: image uint8 or float32 [H,W,3] (numpy or
CUDA tensor), label uint8 [H,W]; returns CUDA tensors — float [3,h,w] + int64 [h,w] after ToTensor, else float32
[h,w,3] + uint8 [h,w].  There is no CPU path: without the HIP library this module raises.
"""
import collections.abc
import ctypes
import math
import numbers
import random

import numpy as np
import torch

from . import ops
from ._lib import lib

MAX_MAPS = 6
K_RESIZE, K_ROTATE, K_BLUR, K_GATHER = 1, 2, 3, 4     # SEMSEG_AUG_* (0 = no op in this round)


class AugMap(ctypes.Structure):
    _fields_ = [("in_h", ctypes.c_int), ("in_w", ctypes.c_int), ("sy", ctypes.c_int), ("oy", ctypes.c_int),
                ("sx", ctypes.c_int), ("ox", ctypes.c_int), ("swap_rb", ctypes.c_int), ("pad_lab", ctypes.c_int),
                ("pad", ctypes.c_float * 3), ("reserved", ctypes.c_int)]


class AugOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("src_u8", ctypes.c_int),
                ("src_img", ctypes.c_ulonglong), ("src_lab", ctypes.c_ulonglong),
                ("dst_img", ctypes.c_ulonglong), ("dst_lab", ctypes.c_ulonglong),
                ("src_H", ctypes.c_int), ("src_W", ctypes.c_int), ("src_y0", ctypes.c_int), ("src_x0", ctypes.c_int),
                ("src_h", ctypes.c_int), ("src_w", ctypes.c_int),
                ("dst_H", ctypes.c_int), ("dst_W", ctypes.c_int), ("dst_y0", ctypes.c_int), ("dst_x0", ctypes.c_int),
                ("dst_h", ctypes.c_int), ("dst_w", ctypes.c_int),
                ("p", ctypes.c_double * 6), ("pad", ctypes.c_float * 3), ("pad_lab", ctypes.c_int),
                ("ksize", ctypes.c_int), ("n_maps", ctypes.c_int), ("out_chw", ctypes.c_int),
                ("normalize", ctypes.c_int), ("mean", ctypes.c_float * 3), ("std", ctypes.c_float * 3),
                ("reserved", ctypes.c_int * 2), ("maps", AugMap * MAX_MAPS)]


def _is_pair(v):
    return isinstance(v, collections.abc.Iterable) and len(v) == 2


# ------------------------------------------------------------------------------------------------------------------
# plan
# ------------------------------------------------------------------------------------------------------------------
class _Plan:
    """One sample's chain after the parameters are drawn: `items` in application order."""

    def __init__(self, h, w):
        self.h, self.w = int(h), int(w)
        self.items = []
        self.tensor = False
        self.norm = None     # (mean, std | None)

    def geometry(self, what):
        if self.tensor:
            raise RuntimeError("segtransform.%s needs an H x W x C image; only Normalize may follow ToTensor\n" % what)

    def add_map(self, out_h, out_w, sy=1, oy=0, sx=1, ox=0, swap=0, pad=None, pad_lab=0):
        self.items.append(dict(k="map", in_h=self.h, in_w=self.w, out_h=out_h, out_w=out_w, sy=sy, oy=oy, sx=sx,
                               ox=ox, swap=swap, pad=[0.0, 0.0, 0.0] if pad is None else [float(v) for v in pad],
                               pad_lab=int(pad_lab)))
        self.h, self.w = out_h, out_w


def _rotation_inverse(cx, cy, angle_deg):
    """getRotationMatrix2D((cx, cy), angle, 1) followed by warpAffine's own inversion: destination -> source."""
    cx, cy = float(np.float32(cx)), float(np.float32(cy))
    a = angle_deg * math.pi / 180.0
    al, be = math.cos(a), math.sin(a)
    m = [al, be, (1 - al) * cx - be * cy, -be, al, be * cx + (1 - al) * cy]
    det = m[0] * m[4] - m[1] * m[3]
    det = 1.0 / det if det != 0 else 0.0
    i0, i4 = m[4] * det, m[0] * det
    i1, i3 = m[1] * -det, m[3] * -det
    return [i0, i1, -i0 * m[2] - i1 * m[5], i3, i4, -i3 * m[2] - i4 * m[5]]


# ------------------------------------------------------------------------------------------------------------------
# the reference's classes (util/transform.py), as planners
# ------------------------------------------------------------------------------------------------------------------
class ToTensor(object):
    """transform.py:22-41: HWC image -> float CHW, label -> int64 (fused into the final gather)."""

    def plan(self, st):
        st.geometry("ToTensor()")
        st.tensor = True


class Normalize(object):
    """transform.py:44-61: channel = (channel - mean) / std on the CHW tensor."""

    def __init__(self, mean, std=None):
        if std is None:
            assert len(mean) > 0
        else:
            assert len(mean) == len(std)
        self.mean, self.std = mean, std

    def plan(self, st):
        if not st.tensor or st.norm is not None:
            raise RuntimeError("segtransform.Normalize() works on the tensor ToTensor() returns, once\n")
        if len(self.mean) != 3:
            raise RuntimeError("segtransform.Normalize() on the device handles 3-channel images\n")
        st.norm = (list(self.mean), None if self.std is None else list(self.std))


class Resize(object):
    """transform.py:64-73: size = (h, w); image INTER_LINEAR, label INTER_NEAREST."""

    def __init__(self, size):
        assert _is_pair(size)
        self.size = size

    def plan(self, st):
        st.geometry("Resize()")
        dh, dw = int(self.size[0]), int(self.size[1])
        st.items.append(dict(k="resize", in_h=st.h, in_w=st.w, out_h=dh, out_w=dw,
                             scale_x=1.0 / (dw / st.w), scale_y=1.0 / (dh / st.h)))
        st.h, st.w = dh, dw


class RandScale(object):
    """transform.py:76-103: scale in [scale_min, scale_max], optional aspect-ratio jitter."""

    def __init__(self, scale, aspect_ratio=None):
        assert _is_pair(scale)
        if isinstance(scale[0], numbers.Number) and isinstance(scale[1], numbers.Number) and 0 < scale[0] < scale[1]:
            self.scale = scale
        else:
            raise RuntimeError("segtransform.RandScale() scale param error.\n")
        if aspect_ratio is None:
            self.aspect_ratio = None
        elif _is_pair(aspect_ratio) and isinstance(aspect_ratio[0], numbers.Number) \
                and isinstance(aspect_ratio[1], numbers.Number) and 0 < aspect_ratio[0] < aspect_ratio[1]:
            self.aspect_ratio = aspect_ratio
        else:
            raise RuntimeError("segtransform.RandScale() aspect_ratio param error.\n")

    def plan(self, st):
        st.geometry("RandScale()")
        rng = st.rng
        s = self.scale[0] + (self.scale[1] - self.scale[0]) * rng.random()
        ar = 1.0
        if self.aspect_ratio is not None:
            ar = math.sqrt(self.aspect_ratio[0] + (self.aspect_ratio[1] - self.aspect_ratio[0]) * rng.random())
        fx, fy = s * ar, s / ar
        dw, dh = int(round(st.w * fx)), int(round(st.h * fy))      # saturate_cast<int>: round half to even
        if dw <= 0 or dh <= 0:
            raise RuntimeError("segtransform.RandScale() produced an empty image\n")
        st.items.append(dict(k="resize", in_h=st.h, in_w=st.w, out_h=dh, out_w=dw, scale_x=1.0 / fx, scale_y=1.0 / fy))
        st.h, st.w = dh, dw


class Crop(object):
    """transform.py:106-164: pad to at least (crop_h, crop_w) with `padding` / `ignore_label`, then a random or
    centred window."""

    def __init__(self, size, crop_type='center', padding=None, ignore_label=255):
        if isinstance(size, int):
            self.crop_h = self.crop_w = size
        elif _is_pair(size) and isinstance(size[0], int) and isinstance(size[1], int) and size[0] > 0 and size[1] > 0:
            self.crop_h, self.crop_w = size[0], size[1]
        else:
            raise RuntimeError("crop size error.\n")
        if crop_type not in ('center', 'rand'):
            raise RuntimeError("crop type error: rand | center\n")
        self.crop_type = crop_type
        if padding is None:
            self.padding = None
        elif isinstance(padding, list):
            if not all(isinstance(i, numbers.Number) for i in padding):
                raise RuntimeError("padding in Crop() should be a number list\n")
            if len(padding) != 3:
                raise RuntimeError("padding channel is not equal with 3\n")
            self.padding = padding
        else:
            raise RuntimeError("padding in Crop() should be a number list\n")
        if not isinstance(ignore_label, int):
            raise RuntimeError("ignore_label should be an integer number\n")
        self.ignore_label = ignore_label

    def plan(self, st):
        st.geometry("Crop()")
        pad_h, pad_w = max(self.crop_h - st.h, 0), max(self.crop_w - st.w, 0)
        top, left = int(pad_h / 2), int(pad_w / 2)
        if (pad_h > 0 or pad_w > 0) and self.padding is None:
            raise RuntimeError("segtransform.Crop() need padding while padding argument is None\n")
        h, w = st.h + pad_h, st.w + pad_w
        if self.crop_type == 'rand':
            y0 = st.rng.randint(0, h - self.crop_h)
            x0 = st.rng.randint(0, w - self.crop_w)
        else:
            y0, x0 = int((h - self.crop_h) / 2), int((w - self.crop_w) / 2)
        st.add_map(self.crop_h, self.crop_w, oy=y0 - top, ox=x0 - left, pad=self.padding, pad_lab=self.ignore_label)


class RandRotate(object):
    """transform.py:167-195: with probability p rotate about the centre by an angle in [rotate_min, rotate_max]."""

    def __init__(self, rotate, padding, ignore_label=255, p=0.5):
        assert _is_pair(rotate)
        if isinstance(rotate[0], numbers.Number) and isinstance(rotate[1], numbers.Number) and rotate[0] < rotate[1]:
            self.rotate = rotate
        else:
            raise RuntimeError("segtransform.RandRotate() scale param error.\n")
        assert padding is not None
        assert isinstance(padding, list) and len(padding) == 3
        if not all(isinstance(i, numbers.Number) for i in padding):
            raise RuntimeError("padding in RandRotate() should be a number list\n")
        self.padding = padding
        assert isinstance(ignore_label, int)
        self.ignore_label = ignore_label
        self.p = p

    def plan(self, st):
        st.geometry("RandRotate()")
        if st.rng.random() < self.p:
            angle = self.rotate[0] + (self.rotate[1] - self.rotate[0]) * st.rng.random()
            st.items.append(dict(k="rotate", in_h=st.h, in_w=st.w, out_h=st.h, out_w=st.w,
                                 m=_rotation_inverse(st.w / 2, st.h / 2, angle),
                                 pad=[float(v) for v in self.padding], pad_lab=self.ignore_label))


class RandomHorizontalFlip(object):
    """transform.py:198-206"""

    def __init__(self, p=0.5):
        self.p = p

    def plan(self, st):
        st.geometry("RandomHorizontalFlip()")
        if st.rng.random() < self.p:
            st.add_map(st.h, st.w, sx=-1, ox=st.w - 1)


class RandomVerticalFlip(object):
    """transform.py:209-217"""

    def __init__(self, p=0.5):
        self.p = p

    def plan(self, st):
        st.geometry("RandomVerticalFlip()")
        if st.rng.random() < self.p:
            st.add_map(st.h, st.w, sy=-1, oy=st.h - 1)


class RandomGaussianBlur(object):
    """transform.py:220-227: with probability 0.5, GaussianBlur((radius, radius), sigma 0) on the image only."""

    def __init__(self, radius=5):
        self.radius = radius

    def plan(self, st):
        st.geometry("RandomGaussianBlur()")
        if st.rng.random() < 0.5:
            if self.radius not in (1, 3, 5, 7):
                raise NotImplementedError("sigma-0 Gaussian kernels are tabulated for radius 1, 3, 5, 7")
            if self.radius > 1:
                st.items.append(dict(k="blur", in_h=st.h, in_w=st.w, out_h=st.h, out_w=st.w, ksize=self.radius))


class RGB2BGR(object):
    """transform.py:230-234"""

    def plan(self, st):
        st.geometry("RGB2BGR()")
        st.add_map(st.h, st.w, swap=1)


class BGR2RGB(RGB2BGR):
    """transform.py:237-241"""


# ------------------------------------------------------------------------------------------------------------------
# region propagation
# ------------------------------------------------------------------------------------------------------------------
def _lin_src(d, scale, n):
    f = np.float32((d + 0.5) * scale - 0.5)
    s = int(np.floor(f))
    if s < 0:
        s = 0
    if s >= n - 1:
        s = n - 1
    return s, min(s + 1, n - 1)


def _resize_need(a, b, scale, n):
    """source index range read by destination indices a..b (linear for the image, nearest for the label)."""
    lo, hi = _lin_src(a, scale, n)[0], _lin_src(b, scale, n)[1]
    nlo, nhi = min(int(math.floor(a * scale)), n - 1), min(int(math.floor(b * scale)), n - 1)
    return min(lo, nlo), max(hi, nhi)


def _need(stage, roi):
    """roi = (y0, x0, h, w) of the stage's output that is needed -> region of its input that it reads."""
    y0, x0, h, w = roi
    if h <= 0 or w <= 0:
        return (0, 0, 0, 0)
    ya, yb, xa, xb = y0, y0 + h - 1, x0, x0 + w - 1
    H, W = stage["in_h"], stage["in_w"]
    k = stage["k"]
    if k == "resize":
        ya, yb = _resize_need(ya, yb, stage["scale_y"], H)
        xa, xb = _resize_need(xa, xb, stage["scale_x"], W)
    elif k == "blur":
        r = stage["ksize"] // 2
        ya, yb, xa, xb = max(ya - r, 0), min(yb + r, H - 1), max(xa - r, 0), min(xb + r, W - 1)
    elif k == "rotate":
        m = stage["m"]
        xs = [m[0] * x + m[1] * y + m[2] for x in (xa, xb) for y in (ya, yb)]
        ys = [m[3] * x + m[4] * y + m[5] for x in (xa, xb) for y in (ya, yb)]
        xa, xb = max(int(math.floor(min(xs))) - 2, 0), min(int(math.ceil(max(xs))) + 2, W - 1)
        ya, yb = max(int(math.floor(min(ys))) - 2, 0), min(int(math.ceil(max(ys))) + 2, H - 1)
    else:  # gather: through the maps, last to first
        for mp in reversed(stage["maps"]):
            ya, yb = sorted((mp["sy"] * ya + mp["oy"], mp["sy"] * yb + mp["oy"]))
            xa, xb = sorted((mp["sx"] * xa + mp["ox"], mp["sx"] * xb + mp["ox"]))
            ya, yb, xa, xb = max(ya, 0), min(yb, mp["in_h"] - 1), max(xa, 0), min(xb, mp["in_w"] - 1)
            if ya > yb or xa > xb:
                return (0, 0, 0, 0)
    if ya > yb or xa > xb:
        return (0, 0, 0, 0)
    return (ya, xa, yb - ya + 1, xb - xa + 1)


def _stages(plan, in_h, in_w):
    """Group the plan into launchable stages: resampling ops as they are, runs of index maps as one gather."""
    out, maps = [], []
    h, w = in_h, in_w

    def flush(final):
        nonlocal maps
        for i in range(0, max(len(maps), 1), MAX_MAPS):      # > MAX_MAPS maps in a row: several gathers
            chunk = maps[i:i + MAX_MAPS]
            ih, iw = (chunk[0]["in_h"], chunk[0]["in_w"]) if chunk else (h, w)
            oh, ow = (chunk[-1]["out_h"], chunk[-1]["out_w"]) if chunk else (h, w)
            last = final and i + MAX_MAPS >= len(maps)
            out.append(dict(k="gather", in_h=ih, in_w=iw, out_h=oh, out_w=ow, maps=chunk, final=last))
        maps = []

    for it in plan.items:
        if it["k"] == "map":
            maps.append(it)
        else:
            if maps:
                flush(False)
            out.append(dict(it, final=False))
        h, w = it["out_h"], it["out_w"]
    if plan.tensor or maps or not out:
        flush(True)
    out[-1]["final"] = True
    return out


# ------------------------------------------------------------------------------------------------------------------
# execution
# ------------------------------------------------------------------------------------------------------------------
def _align(n, a=256):
    return (n + a - 1) // a * a


def _as_source(image, label):
    """-> (image is uint8, h, w) after the shape/dtype checks (the reference's are in ToTensor, transform.py:24-34)."""
    ishape = tuple(image.shape)
    if len(ishape) != 3 or ishape[2] != 3:
        raise RuntimeError("segtransform on the device handles H x W x 3 images (as read by SemData)\n")
    if tuple(label.shape) != ishape[:2]:
        raise RuntimeError("Image & label shape mismatch\n")
    idt = str(image.dtype).replace("torch.", "")
    ldt = str(label.dtype).replace("torch.", "")
    if idt not in ("uint8", "float32") or ldt != "uint8":
        raise RuntimeError("segtransform expects a uint8 or float32 image and a uint8 label\n")
    return idt == "uint8", ishape[0], ishape[1]


class Compose(object):
    """transform.py:11-19.  `rng`: object with random()/randint() (default: the `random` module, as the reference)."""

    def __init__(self, segtransform, rng=None):
        self.segtransform = segtransform
        self.rng = rng if rng is not None else random
        if ctypes.sizeof(AugOp) != lib.semseg_aug_op_size():
            raise RuntimeError("semseg_aug_op layout mismatch between include/semseg_hip.h and transform.py")

    def plan(self, h, w):
        st = _Plan(h, w)
        st.rng = self.rng
        for t in self.segtransform:
            t.plan(st)
        return st

    def schedule(self, sizes):
        """Host-only part of `batch`: draw every sample's parameters (sample order = the reference's draw order), group
        the plan into stages and propagate the needed regions from the last stage to the first.  -> (plans, chains);
        each stage dict carries dst_roi (region it writes) and src_need (region of its input it reads)."""
        plans = [self.plan(h, w) for (h, w) in sizes]
        chains = [_stages(p, h, w) for p, (h, w) in zip(plans, sizes)]
        for ch in chains:
            roi = (0, 0, ch[-1]["out_h"], ch[-1]["out_w"])
            for stg in reversed(ch):
                stg["dst_roi"] = roi
                roi = _need(stg, roi)
                stg["src_need"] = roi
        return plans, chains

    def __call__(self, image, label):
        imgs, labs = self.batch([image], [label], stack=False)
        return imgs[0], labs[0]

    def batch(self, images, labels, stack=True, device=None, return_plans=False):
        """Transform a list of samples.  With `stack` and equal output sizes: ([B,3,h,w] float, [B,h,w] int64) for
        chains ending in ToTensor; otherwise lists of per-sample tensors."""
        assert len(images) == len(labels) and len(images) > 0
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        B = len(images)
        metas = [_as_source(im, lb) for im, lb in zip(images, labels)]
        plans, chains = self.schedule([(h, w) for (_, h, w) in metas])

        # one arena: host-resident sources first (one H2D copy), then the intermediate regions
        off = 0
        host_parts = []
        src_ptrs = []
        for b, (im, lb) in enumerate(zip(images, labels)):
            if isinstance(im, torch.Tensor) and im.is_cuda:
                assert lb.is_cuda and im.is_contiguous() and lb.is_contiguous()
                src_ptrs.append((im.data_ptr(), lb.data_ptr()))
            else:
                ia = np.ascontiguousarray(im.cpu().numpy() if isinstance(im, torch.Tensor) else im)
                la = np.ascontiguousarray(lb.cpu().numpy() if isinstance(lb, torch.Tensor) else lb)
                io = off
                off = _align(off + ia.nbytes)
                lo = off
                off = _align(off + la.nbytes)
                host_parts.append((io, ia.view(np.uint8).reshape(-1), lo, la.reshape(-1)))
                src_ptrs.append((-1 - io, -1 - lo))                    # arena-relative, patched below
        host_bytes = off
        uniform = stack and all(p.tensor for p in plans) and \
            len({(ch[-1]["out_h"], ch[-1]["out_w"]) for ch in chains}) == 1
        for ch, plan in zip(chains, plans):
            for stg in ch:
                _, _, h, w = stg["dst_roi"]
                stg["chw"] = stg["final"] and stg["k"] == "gather" and plan.tensor
                if stg["chw"] and uniform:
                    stg["off"] = None                                   # written into the batch tensors
                    continue
                stg["off"] = (off, _align(off + h * w * 12))
                off = _align(stg["off"][1] + h * w * (8 if stg["chw"] else 1))
        arena = torch.empty(max(off, 256), dtype=torch.uint8, device=dev)
        if host_parts:
            staging = np.empty(host_bytes, dtype=np.uint8)
            for io, ia, lo, la in host_parts:
                staging[io:io + ia.size] = ia
                staging[lo:lo + la.size] = la
            arena[:host_bytes].copy_(torch.from_numpy(staging), non_blocking=False)
        base = arena.data_ptr()
        src_ptrs = [(base + (-1 - a) if a < 0 else a, base + (-1 - l) if l < 0 else l) for a, l in src_ptrs]

        out_img = out_lab = None
        if uniform:
            oh, ow = chains[0][-1]["out_h"], chains[0][-1]["out_w"]
            out_img = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=dev)
            out_lab = torch.empty(B, oh, ow, dtype=torch.int64, device=dev)

        # descriptors: [round][sample]
        R = max(len(ch) for ch in chains)
        table = (AugOp * (R * B))()
        max_pix = [0] * R
        results = []
        for b, (ch, plan, (is_u8, h0, w0)) in enumerate(zip(chains, plans, metas)):
            view = dict(img=src_ptrs[b][0], lab=src_ptrs[b][1], u8=is_u8, H=h0, W=w0, roi=(0, 0, h0, w0))
            for r, stg in enumerate(ch):
                o = table[r * B + b]
                y0, x0, h, w = stg["dst_roi"]
                o.kind = {"resize": K_RESIZE, "rotate": K_ROTATE, "blur": K_BLUR, "gather": K_GATHER}[stg["k"]]
                o.src_u8 = 1 if view["u8"] else 0
                o.src_img, o.src_lab = view["img"], view["lab"]
                o.src_H, o.src_W = view["H"], view["W"]
                o.src_y0, o.src_x0, o.src_h, o.src_w = view["roi"]
                ny0, nx0, nh, nw = stg["src_need"]
                vy0, vx0, vh, vw = view["roi"]
                assert nh == 0 or (ny0 >= vy0 and nx0 >= vx0 and ny0 + nh <= vy0 + vh and nx0 + nw <= vx0 + vw), \
                    "planner: stage reads outside the region its producer wrote"
                o.dst_H, o.dst_W = stg["out_h"], stg["out_w"]
                o.dst_y0, o.dst_x0, o.dst_h, o.dst_w = y0, x0, h, w
                if stg["off"] is None:
                    o.dst_img, o.dst_lab = out_img[b].data_ptr(), out_lab[b].data_ptr()
                else:
                    o.dst_img, o.dst_lab = base + stg["off"][0], base + stg["off"][1]
                if stg["k"] == "resize":
                    o.p[0], o.p[1] = stg["scale_x"], stg["scale_y"]
                elif stg["k"] == "rotate":
                    for i in range(6):
                        o.p[i] = stg["m"][i]
                    for i in range(3):
                        o.pad[i] = stg["pad"][i]
                    o.pad_lab = min(max(int(stg["pad_lab"]), 0), 255)
                elif stg["k"] == "blur":
                    o.ksize = stg["ksize"]
                else:
                    o.n_maps = len(stg["maps"])
                    for i, mp in enumerate(stg["maps"]):
                        m = o.maps[i]
                        m.in_h, m.in_w, m.sy, m.oy, m.sx, m.ox = mp["in_h"], mp["in_w"], mp["sy"], mp["oy"], mp["sx"], mp["ox"]
                        m.swap_rb, m.pad_lab = mp["swap"], min(max(mp["pad_lab"], 0), 255)
                        for c in range(3):
                            m.pad[c] = mp["pad"][c]
                    if stg["chw"]:
                        o.out_chw = 1
                        if plan.norm is not None:
                            mean, std = plan.norm
                            o.normalize = 1 if std is None else 2
                            for c in range(3):
                                o.mean[c] = mean[c]
                                o.std[c] = 1.0 if std is None else std[c]
                max_pix[r] = max(max_pix[r], h * w)
                view = dict(img=o.dst_img, lab=o.dst_lab, u8=False, H=stg["out_h"], W=stg["out_w"], roi=stg["dst_roi"])
            if not uniform:
                last = ch[-1]
                _, _, h, w = last["dst_roi"]
                a0, a1 = last["off"]
                if plan.tensor:
                    results.append((arena[a0:a0 + h * w * 12].view(torch.float32).view(3, h, w),
                                    arena[a1:a1 + h * w * 8].view(torch.int64).view(h, w)))
                else:
                    results.append((arena[a0:a0 + h * w * 12].view(torch.float32).view(h, w, 3),
                                    arena[a1:a1 + h * w].view(h, w)))

        sz = ctypes.sizeof(AugOp)
        ops_dev = torch.from_numpy(np.frombuffer(table, dtype=np.uint8).copy()).to(dev)
        for r in range(R):
            ops.augment_round(ops_dev[r * B * sz:], B, max_pix[r])
        if uniform:
            res = (out_img, out_lab)
        else:
            res = ([a for a, _ in results], [l for _, l in results])
        return res + (plans,) if return_plans else res
