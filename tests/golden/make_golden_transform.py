"""Generates tests/golden/transform_ref.npz: the REAL reference transform classes (/root/reference/util/transform.py,
imported here) run on the cases of tests/transform_cases.py with `cv2` replaced by oracle/cv2_restated.py (OpenCV is
not installed and not vendored).  This pins the draw order from `random`, the parameter formulas, Crop's pad/offset
logic and ToTensor/Normalize to the reference bit for bit; the cv2 primitives stay "parity unpinned".
Also asserts oracle/transform.py reproduces every case exactly.  Build container only:
    python tests/golden/make_golden_transform.py
"""
import collections
import collections.abc
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import cv2_restated, transform as otf   # noqa: E402
import transform_cases as tc                         # noqa: E402


def import_reference_transform():
    stub = types.ModuleType("cv2")
    for k in dir(cv2_restated):
        if not k.startswith("_"):
            setattr(stub, k, getattr(cv2_restated, k))
    sys.modules["cv2"] = stub
    collections.Iterable = collections.abc.Iterable      # the reference predates python 3.10
    sys.path.insert(0, REF)
    import util.transform as rt
    sys.path.remove(REF)
    del sys.modules["cv2"]
    return rt


def to_np(v):
    return v.numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v)


def main():
    rt = import_reference_transform()
    blob = {}
    for name, (H, W, ops, seeds) in tc.CASES.items():
        img_u8, lab = tc.make_input(name, H, W)
        chain = tc.build_chain(rt, ops)
        for seed in seeds:
            random.seed(seed)
            ri, rl = chain(np.float32(img_u8), lab.copy())
            random.seed(seed)
            oi, ol = otf.run(ops, np.float32(img_u8), lab.copy())
            ri, rl, oi, ol = to_np(ri), to_np(rl), to_np(oi), to_np(ol)
            assert ri.dtype == oi.dtype and rl.dtype == ol.dtype, (name, ri.dtype, oi.dtype, rl.dtype, ol.dtype)
            assert ri.shape == oi.shape and np.array_equal(ri, oi), "oracle != reference: image %s/%d" % (name, seed)
            assert np.array_equal(rl, ol), "oracle != reference: label %s/%d" % (name, seed)
            blob["%s/%d/image" % (name, seed)] = ri
            blob["%s/%d/label" % (name, seed)] = rl.astype(np.uint8) if rl.dtype == np.int64 else rl
            print("%-22s seed %2d -> image %s %s label %s" % (name, seed, ri.shape, ri.dtype, rl.shape))
    path = os.path.join(HERE, "transform_ref.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
