"""The reference's dataset front end (util/dataset.py) for the device-side input pipeline.

`make_dataset` parses the list file exactly as util/dataset.py:17-49 does.  `SemData` decodes on the CPU worker and
stops there: it returns the decoded uint8 arrays (image RGB [H,W,3], label [H,W]) instead of running cv2 transforms
on float32 copies (dataset.py:56-70).

Who runs what (the DataLoader calls `collate_fn` INSIDE its worker processes when num_workers > 0, and a forked worker
must never touch HIP):
  * workers (tool/train.py:202-207 runs 16 of them): file decode only; `raw_collate` passes the list of uint8 pairs
    through untouched;
  * training process: `DeviceLoader(loader, compose)` wraps the DataLoader; every raw batch it yields goes through
    semseg_amd.transform.Compose.batch on the GPU and comes out as the [B,3,h,w] float / [B,h,w] int64 CUDA tensors the
    train step consumes (no pinned-memory float batch, no per-sample ToTensor/Normalize on the host).
`DeviceCollate(compose)` is the num_workers == 0 shortcut (collate_fn in the training process); it raises inside a
worker, and so does a `SemData(transform=...)` whose transform would run there.

Decode: OpenCV is not installed in this image, so files are decoded with PIL.  For lossless files (PNG/BMP/PPM/PGM)
RGB and 8-bit grey decode to the same bytes `cv2.imread(IMREAD_COLOR)` + BGR2RGB / `IMREAD_GRAYSCALE` produce; JPEG
bytes depend on the decoder build (both normally libjpeg-turbo) — not pinned here.  Labels must be 8-bit grey files
(what the reference's dataset lists point at); anything that would need a colour->grey conversion is rejected
rather than converted with a different formula than cv2's.
"""
import os

import numpy as np
from torch.utils.data import Dataset, get_worker_info

_WORKER_MSG = ("%s would run the HIP transform chain inside a DataLoader worker process (num_workers > 0): a forked "
               "worker cannot initialise the GPU.  Keep the workers decode-only: DataLoader(SemData(...), "
               "collate_fn=raw_collate, num_workers=N) wrapped in DeviceLoader(loader, compose)\n")

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
            if get_worker_info() is not None:
                raise RuntimeError(_WORKER_MSG % "SemData(transform=...)")
            image, label = self.transform(image, label)
        return image, label


def raw_collate(samples):
    """collate_fn for decode-only workers: the list of (uint8 image, uint8 label) pairs, untouched (images of one batch
    have different sizes before the chain's Crop, so there is nothing to stack yet)."""
    return [(s[0], s[1]) for s in samples]


class DeviceCollate(object):
    """collate_fn for num_workers == 0 only: list of (uint8 image, uint8 label) -> (input [B,3,h,w] float32, target
    [B,h,w] int64) on the GPU, through tool/train.py:194-201's chain built from semseg_amd.transform classes.  With
    worker processes use raw_collate + DeviceLoader instead (this raises inside a worker)."""

    def __init__(self, compose, device=None):
        self.compose = compose
        self.device = device

    def __call__(self, samples):
        if get_worker_info() is not None:
            raise RuntimeError(_WORKER_MSG % "DeviceCollate")
        images = [s[0] for s in samples]
        labels = [s[1] for s in samples]
        x, y = self.compose.batch(images, labels, stack=True, device=self.device)
        if isinstance(x, list):
            raise RuntimeError("DeviceCollate: the chain must end in ToTensor() and give every sample the same size "
                               "(a Crop), as tool/train.py:194-201 does; got per-sample outputs\n")
        return x, y


class DeviceLoader(object):
    """Iterable over a DataLoader whose workers only decode (collate_fn=raw_collate): the device-side chain runs HERE,
    in the training process, when a batch is taken.  Keeps the loader surface tool/train.py uses: `len(loader)`
    (train.py:252,297), iteration (train.py:264), `.sampler` / `.dataset` / `.batch_size`."""

    def __init__(self, loader, compose, device=None):
        if getattr(loader, "collate_fn", None) is not raw_collate:
            raise RuntimeError("DeviceLoader: build the DataLoader with collate_fn=semseg_amd.dataset.raw_collate "
                               "(workers decode, the transform chain runs in the training process)\n")
        self.loader = loader
        self.collate = DeviceCollate(compose, device)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for raw in self.loader:
            yield self.collate(raw)

    def __getattr__(self, name):
        # only reached for attributes this object does not have: forward plain attributes to the DataLoader, but never
        # dunder probes (copy / pickle protocols) and never before `loader` exists (that lookup would recurse)
        if name.startswith("__") or "loader" not in self.__dict__:
            raise AttributeError(name)
        return getattr(self.loader, name)
