"""The reference's dataset front end (util/dataset.py) for the device-side input pipeline.

`make_dataset` parses the list file exactly as util/dataset.py:17-49 does.  `SemData` decodes on the CPU worker and
stops there: it returns the decoded uint8 arrays (image RGB [H,W,3], label [H,W]) instead of running cv2 transforms
on float32 copies (dataset.py:56-70).  `DeviceCollate(compose)` is the DataLoader `collate_fn` that finishes the
job in the training process: the whole batch goes through semseg_amd.transform.Compose.batch on the GPU and comes
out as the [B,3,h,w] float / [B,h,w] int64 CUDA tensors the train step consumes (no pinned-memory float batch, no
per-sample ToTensor/Normalize on the host).

Decode: OpenCV is not installed in this image, so files are decoded with PIL.  For lossless files (PNG/BMP/PPM/PGM)
RGB and 8-bit grey decode to the same bytes `cv2.imread(IMREAD_COLOR)` + BGR2RGB / `IMREAD_GRAYSCALE` produce; JPEG
bytes depend on the decoder build (both normally libjpeg-turbo) — not pinned here.  Labels must be 8-bit grey files
(what the reference's dataset lists point at); anything that would need a colour->grey conversion is rejected
rather than converted with a different formula than cv2's.
"""
import os

import numpy as np
from torch.utils.data import Dataset

def make_dataset(split='train', data_root=None, data_list=None):
    """util/dataset.py:17-49: 'image label' pairs per line (one path per line for split == 'test')."""
    assert split in ('train', 'val', 'test')
    if not os.path.isfile(data_list):
        raise RuntimeError("no such image list file: %s\n" % data_list)
    columns = 1 if split == 'test' else 2
    pairs = []
    with open(data_list) as f:
        for raw in f:
            fields = raw.strip().split(' ')
            if len(fields) != columns:
                raise RuntimeError("image list line does not have %d column(s): %r\n" % (columns, raw.strip()))
            image_name = os.path.join(data_root, fields[0])
            # test split: the label slot repeats the image path as a placeholder (dataset.py:33)
            pairs.append((image_name, os.path.join(data_root, fields[1]) if columns == 2 else image_name))
    return pairs


def read_image_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def read_label_grey(path):
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "RGB" or im.mode == "RGBA" or im.mode == "P":
            # test split: the label slot holds the image path as a placeholder (dataset.py:33), never used
            return None
        if im.mode != "L":
            raise RuntimeError("label file is not 8-bit grey: " + path + "\n")
        return np.asarray(im, dtype=np.uint8)


class SemData(Dataset):
    """util/dataset.py:52-70.  `transform`, if given, is applied per sample on the device (reference call shape
    `transform(image, label)`); for training leave it None and give the chain to `DeviceCollate` instead."""

    def __init__(self, split='train', data_root=None, data_list=None, transform=None):
        self.split = split
        self.data_list = make_dataset(split, data_root, data_list)
        self.transform = transform

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, index):
        image_path, label_path = self.data_list[index]
        image = read_image_rgb(image_path)
        label = read_label_grey(label_path)
        if label is None:
            if self.split != 'test':
                raise RuntimeError("label file is not 8-bit grey: " + label_path + "\n")
            label = np.zeros(image.shape[:2], dtype=np.uint8)
        if image.shape[0] != label.shape[0] or image.shape[1] != label.shape[1]:
            raise RuntimeError("image and label sizes differ: %s %s\n" % (image_path, label_path))
        if self.transform is not None:
            image, label = self.transform(image, label)
        return image, label


class DeviceCollate(object):
    """collate_fn: list of (uint8 image, uint8 label) -> (input [B,3,h,w] float32, target [B,h,w] int64) on the GPU,
    through tool/train.py:194-201's chain built from semseg_amd.transform classes."""

    def __init__(self, compose, device=None):
        self.compose = compose
        self.device = device

    def __call__(self, samples):
        images = [s[0] for s in samples]
        labels = [s[1] for s in samples]
        x, y = self.compose.batch(images, labels, stack=True, device=self.device)
        if isinstance(x, list):
            raise RuntimeError("DeviceCollate: the chain must end in ToTensor() and give every sample the same size "
                               "(a Crop), as tool/train.py:194-201 does; got per-sample outputs\n")
        return x, y
