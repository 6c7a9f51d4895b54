"""Shared case list for the input-pipeline parity tests (tests/golden/make_golden_transform.py writes the
reference outputs of exactly these cases; test_transform_cpu.py / test_transform_gpu.py read them back)."""
import numpy as np

VALUE_SCALE = 255
MEAN = [0.485 * VALUE_SCALE, 0.456 * VALUE_SCALE, 0.406 * VALUE_SCALE]   # tool/train.py:188-190
STD = [0.229 * VALUE_SCALE, 0.224 * VALUE_SCALE, 0.225 * VALUE_SCALE]    # tool/train.py:191-193


def train_chain(crop, scale=(0.5, 2.0), rotate=(-10, 10), ignore=255, aspect=None):
    """tool/train.py:194-201"""
    return [("rand_scale", scale, aspect), ("rand_rotate", rotate, MEAN, ignore, 0.5), ("rand_blur", 5),
            ("rand_hflip", 0.5), ("crop", crop, "rand", MEAN, ignore), ("to_tensor",), ("normalize", MEAN, STD)]


def val_chain(crop, ignore=255):
    """tool/train.py:209-212"""
    return [("crop", crop, "center", MEAN, ignore), ("to_tensor",), ("normalize", MEAN, STD)]


# name -> (input H, W, ops, [seeds])
CASES = {
    "train_65": (97, 131, train_chain((65, 65)), list(range(8))),
    "train_rect": (83, 61, train_chain((57, 73), scale=(0.6, 1.4), rotate=(-25, 25)), [11, 12, 13, 14]),
    "train_aspect": (70, 90, train_chain((49, 49), aspect=(0.7, 1.5)), [21, 22, 23]),
    "val_pad": (50, 60, val_chain((65, 65)), [0]),
    "val_crop": (97, 131, val_chain((65, 65)), [0]),
    "test_only_tensor": (40, 33, [("to_tensor",)], [0]),
    "resize_flips_bgr": (45, 52, [("resize", (37, 64)), ("rand_vflip", 0.5), ("rand_hflip", 0.5), ("swap_rb",),
                                   ("to_tensor",), ("normalize", MEAN, None)], [1, 2, 3, 4]),
    "rotate_always_blur7": (64, 48, [("rand_rotate", (-45, 45), MEAN, 255, 1.0), ("rand_blur", 7), ("rand_blur", 3),
                                      ("to_tensor",)], [5, 6, 7, 8]),
    "crop_then_scale": (80, 80, [("crop", (60, 50), "rand", MEAN, 255), ("rand_hflip", 1.0),
                                  ("rand_scale", (0.8, 1.3), None), ("to_tensor",)], [31, 32]),
    "no_tensor": (60, 70, [("rand_scale", (0.9, 1.1), None), ("rand_hflip", 1.0), ("crop", (41, 41), "rand", MEAN, 255)],
                  [41, 42]),
}


def make_input(name, H, W):
    """Smooth structure + noise, uint8 like a decoded image; label with a 255 band and class blobs."""
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    base = np.stack([127 + 100 * np.sin(yy / 7.0 + c) * np.cos(xx / 5.0 - c) for c in range(3)], axis=2)
    img = np.clip(base + rng.normal(0, 12, size=(H, W, 3)), 0, 255).astype(np.uint8)
    lab = ((yy // 9 + xx // 11) % 19).astype(np.uint8)
    lab[(yy + xx) % 23 == 0] = 255
    return img, lab


def build_chain(T, ops, **kw):
    """op tuples -> Compose of the classes in module T (semseg_amd.transform or the reference's util.transform)."""
    out = []
    for op in ops:
        k = op[0]
        if k == "rand_scale":
            out.append(T.RandScale(list(op[1]), aspect_ratio=None if op[2] is None else list(op[2])))
        elif k == "rand_rotate":
            out.append(T.RandRotate(list(op[1]), padding=op[2], ignore_label=op[3], p=op[4]))
        elif k == "rand_blur":
            out.append(T.RandomGaussianBlur(op[1]))
        elif k == "rand_hflip":
            out.append(T.RandomHorizontalFlip(op[1]))
        elif k == "rand_vflip":
            out.append(T.RandomVerticalFlip(op[1]))
        elif k == "crop":
            out.append(T.Crop(list(op[1]), crop_type=op[2], padding=op[3], ignore_label=op[4]))
        elif k == "resize":
            out.append(T.Resize(op[1]))
        elif k == "swap_rb":
            out.append(T.RGB2BGR())
        elif k == "to_tensor":
            out.append(T.ToTensor())
        elif k == "normalize":
            out.append(T.Normalize(op[1], op[2]))
        else:
            raise ValueError(k)
    return T.Compose(out, **kw)
