"""TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.

Restatement of the reference's segmentation transforms, /root/reference/util/transform.py, as one interpreter over
a list of op tuples (the product mirrors the reference's class API instead: semseg_amd/transform.py).  The draws
from python's `random` happen in the reference's order, so seeding `random` gives the reference's parameters:

  ("rand_scale", (lo, hi), aspect|None)          RandScale.__call__       transform.py:93-103
  ("rand_rotate", (lo, hi), padding, ignore, p)  RandRotate.__call__      transform.py:188-195
  ("rand_blur", radius)                          RandomGaussianBlur       transform.py:224-227
  ("rand_hflip", p) / ("rand_vflip", p)          Random*Flip              transform.py:202-217
  ("crop", (h, w), "rand"|"center", padding, ignore)   Crop.__call__      transform.py:144-164
  ("resize", (h, w))                             Resize.__call__          transform.py:70-73
  ("swap_rb",)                                   RGB2BGR / BGR2RGB        transform.py:230-241
  ("to_tensor",)                                 ToTensor.__call__        transform.py:24-41
  ("normalize", mean, std|None)                  Normalize.__call__       transform.py:54-61

Pinned by tests/golden/transform_ref.npz: the reference's own classes imported from /root/reference and run with
`cv2` replaced by oracle/cv2_restated.py (tests/golden/make_golden_transform.py); that pins the draw order, the
parameter formulas, the crop/pad logic and ToTensor/Normalize to the reference bit for bit.  The cv2 primitives
themselves are "parity unpinned" (see oracle/cv2_restated.py).
"""
import math
import random

import numpy as np
import torch

from . import cv2_restated as cv2


def run(ops, image, label, rng=random):
    """image float32 [H,W,3], label uint8 [H,W] -> what Compose(ops)(image, label) returns."""
    for op in ops:
        kind = op[0]
        if kind == "rand_scale":
            (lo, hi), aspect = op[1], op[2]
            s = lo + (hi - lo) * rng.random()
            ar = 1.0
            if aspect is not None:
                ar = math.sqrt(aspect[0] + (aspect[1] - aspect[0]) * rng.random())
            fx, fy = s * ar, s / ar
            image = cv2.resize(image, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR)
            label = cv2.resize(label, None, fx=fx, fy=fy, interpolation=cv2.INTER_NEAREST)
        elif kind == "rand_rotate":
            (lo, hi), padding, ignore, p = op[1:5]
            if rng.random() < p:
                angle = lo + (hi - lo) * rng.random()
                h, w = label.shape
                M = cv2.getRotationMatrix2D((w / 2, h / 2), angle, 1)
                image = cv2.warpAffine(image, M, (w, h), flags=cv2.INTER_LINEAR, borderValue=padding)
                label = cv2.warpAffine(label, M, (w, h), flags=cv2.INTER_NEAREST, borderValue=ignore)
        elif kind == "rand_blur":
            if rng.random() < 0.5:
                image = cv2.GaussianBlur(image, (op[1], op[1]), 0)
        elif kind in ("rand_hflip", "rand_vflip"):
            if rng.random() < op[1]:
                code = 1 if kind == "rand_hflip" else 0
                image, label = cv2.flip(image, code), cv2.flip(label, code)
        elif kind == "crop":
            (ch, cw), mode, padding, ignore = op[1:5]
            h, w = label.shape
            ph, pw = max(ch - h, 0), max(cw - w, 0)
            t, l = int(ph / 2), int(pw / 2)
            if ph > 0 or pw > 0:
                if padding is None:
                    raise RuntimeError("segtransform.Crop() need padding while padding argument is None\n")
                image = cv2.copyMakeBorder(image, t, ph - t, l, pw - l, cv2.BORDER_CONSTANT, value=padding)
                label = cv2.copyMakeBorder(label, t, ph - t, l, pw - l, cv2.BORDER_CONSTANT, value=ignore)
            h, w = label.shape
            if mode == "rand":
                y0 = rng.randint(0, h - ch)
                x0 = rng.randint(0, w - cw)
            else:
                y0, x0 = int((h - ch) / 2), int((w - cw) / 2)
            image, label = image[y0:y0 + ch, x0:x0 + cw], label[y0:y0 + ch, x0:x0 + cw]
        elif kind == "resize":
            h, w = op[1]
            image = cv2.resize(image, (w, h), interpolation=cv2.INTER_LINEAR)
            label = cv2.resize(label, (w, h), interpolation=cv2.INTER_NEAREST)
        elif kind == "swap_rb":
            image = cv2.cvtColor(image, cv2.COLOR_BGR2RGB)
        elif kind == "to_tensor":
            image = torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1))).float()
            label = torch.from_numpy(np.ascontiguousarray(label)).long()
        elif kind == "normalize":
            mean, std = op[1], op[2]
            for c in range(len(mean)):
                image[c].sub_(mean[c])
                if std is not None:
                    image[c].div_(std[c])
        else:
            raise ValueError(kind)
    return image, label
