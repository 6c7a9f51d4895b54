"""Multi-scale sliding-window inference on the device — the reference's tool/test.py:122-204
(`net_process`, `scale_process`, `test`; also tool/demo.py:106-189) without the per-crop host round
trip.  Same arithmetic order as the reference: for each scale resize the float image with
cv2.INTER_LINEAR semantics, mean-pad to the crop size, slide crop_h x crop_w windows with stride
ceil(crop*2/3), run [crop, flip(crop)] through the model, softmax, average the flips, accumulate and
divide by the visit count, un-pad, resize back, average over scales, argmax.

Differences (deliberate): all crops of one scale go through the network as ONE batch; probabilities are
accumulated in fp32 on the device (the reference uses float64 numpy canvases on the host).

Multi-GPU (BASELINE.json configs[4], SURVEY.md section 8e "Test path"): one process per GPU; the (scale, crop) units
of an image are sharded over the ranks, each rank accumulates its crops into per-scale canvases, normalises by the
full visit count, resizes back and sums over its scales; ONE reduce of the [classes, h, w] sum to rank 0 finishes
the image.  The reference's own multi-GPU test mode shards whole images instead (tool/test.py:88-93).
"""
import math

import torch

from . import ops


def _round(x):
    return int(round(x))  # python's round, as test.py:194,198,200 uses it


class MultiScaleTester:
    def __init__(self, model, classes, base_size, crop_h, crop_w, scales=(1.0,), mean=None, std=None,
                 stride_rate=2.0 / 3.0, max_batch_crops=16, shard=False, group=None, all_ranks=False):
        value_scale = 255
        self.model = model.eval()
        self.classes = classes
        self.base_size, self.crop_h, self.crop_w = base_size, crop_h, crop_w
        self.scales = tuple(scales)
        self.mean = mean if mean is not None else [0.485 * value_scale, 0.456 * value_scale, 0.406 * value_scale]
        self.std = std if std is not None else [0.229 * value_scale, 0.224 * value_scale, 0.225 * value_scale]
        self.stride_rate = stride_rate
        self.max_batch_crops = max_batch_crops
        self.device = next(model.parameters()).device
        self.shard, self.group, self.all_ranks = shard, group, all_ranks

    # ---- geometry, identical to tool/test.py:150-170 ----
    def crop_grid(self, ori_h, ori_w):
        ch, cw = self.crop_h, self.crop_w
        pad_h, pad_w = max(ch - ori_h, 0), max(cw - ori_w, 0)
        ph, pw = int(pad_h / 2), int(pad_w / 2)
        new_h, new_w = ori_h + pad_h, ori_w + pad_w
        stride_h, stride_w = int(math.ceil(ch * self.stride_rate)), int(math.ceil(cw * self.stride_rate))
        grid_h = int(math.ceil(float(new_h - ch) / stride_h) + 1)
        grid_w = int(math.ceil(float(new_w - cw) / stride_w) + 1)
        pos = []
        for ih in range(grid_h):
            for iw in range(grid_w):
                s_h = ih * stride_h
                e_h = min(s_h + ch, new_h)
                s_h = e_h - ch
                s_w = iw * stride_w
                e_w = min(s_w + cw, new_w)
                s_w = e_w - cw
                pos.append((s_h, s_w))
        return ph, pw, new_h, new_w, pos

    def scaled_size(self, h, w, scale):
        long_size = _round(scale * self.base_size)
        new_h = new_w = long_size
        if h > w:
            new_w = _round(long_size / float(h) * w)
        else:
            new_h = _round(long_size / float(w) * h)
        return new_h, new_w

    def num_forwards(self, h, w):
        """Crops x 2 flips over all scales (46 for a 512x512 image at the six ADE scales)."""
        return sum(2 * len(self.crop_grid(*self.scaled_size(h, w, s))[4]) for s in self.scales)

    # ---- work plan: one unit = one crop (x 2 flips) of one scale ----
    def plan(self, h, w):
        """Per scale: resized size, padding, padded canvas size, crop origins (tool/test.py:150-170,191-201)."""
        out = []
        for scale in self.scales:
            sh, sw = self.scaled_size(h, w, scale)
            ph, pw, new_h, new_w, pos = self.crop_grid(sh, sw)
            out.append(dict(scale=scale, sh=sh, sw=sw, ph=ph, pw=pw, new_h=new_h, new_w=new_w, pos=pos))
        return out

    @staticmethod
    def shard_units(plan, rank, world):
        """BASELINE.json configs[4]: the crops of all scales form one list of independent units; rank r takes the
        r-th contiguous slice (sizes differ by at most one; contiguous keeps a rank's crops on as few scales as
        possible, i.e. larger batches).  Returns {scale index: [crop indices]}.  The reference only shards whole
        images (tool/test.py:88-93 index_start/index_step)."""
        units = [(si, ci) for si, sc in enumerate(plan) for ci in range(len(sc["pos"]))]
        n = len(units)
        lo, hi = rank * n // world, (rank + 1) * n // world
        mine = {}
        for si, ci in units[lo:hi]:
            mine.setdefault(si, []).append(ci)
        return mine

    def _dist(self):
        import torch.distributed as dist
        if self.shard and dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(self.group), dist.get_world_size(self.group)
        return None, 0, 1

    def _new_total(self, C, h, w):
        return torch.zeros(C, h, w, dtype=torch.float32, device=self.device)

    def _accumulate_scale(self, img, h, w, sc, crops, total, sharded):
        """total += (1/len(scales)) * resize_back(canvas / count) for the given crops of one scale.  The visit count
        is the FULL count of the scale (every rank knows the geometry), so the contributions of different ranks
        add up to the single-process result (the resize is linear in the canvas)."""
        dev = self.device
        C, ch, cw = self.classes, self.crop_h, self.crop_w
        sh, sw, ph, pw, new_h, new_w = sc["sh"], sc["sw"], sc["ph"], sc["pw"], sc["new_h"], sc["new_w"]
        scaled = torch.empty(sh, sw, 3, dtype=torch.float32, device=dev)
        ops.resize_linear_hwc(img, h, w, scaled, sh, sw, 3)
        canvas = torch.zeros(C, new_h, new_w, dtype=torch.float32, device=dev)
        count = torch.zeros(new_h, new_w, dtype=torch.float32, device=dev)
        pos = [sc["pos"][i] for i in crops]
        for b0 in range(0, len(pos), self.max_batch_crops):
            chunk = pos[b0:b0 + self.max_batch_crops]
            K = len(chunk)
            pos_dev = torch.tensor(chunk, dtype=torch.int32, device=dev).contiguous()
            org_dev = torch.tensor([(y - ph, x - pw) for y, x in chunk], dtype=torch.int32, device=dev)
            batch = torch.empty(2 * K, 3, ch, cw, dtype=torch.float32, device=dev)
            ops.crop_normalize_flip(scaled, sh, sw, org_dev, K, ch, cw, self.mean, self.std, batch)
            logits = self.model(batch)                      # [2K, C, ch, cw] (zoom_factor 8)
            assert tuple(logits.shape) == (2 * K, C, ch, cw), "model must return crop-sized logits"
            ops.softmax_flip_accumulate(logits.contiguous(), pos_dev, K, C, ch, cw, canvas, count, new_h, new_w)
        if sharded:
            import numpy as np
            full = np.zeros((new_h, new_w), dtype=np.float32)
            for y, x in sc["pos"]:
                full[y:y + ch, x:x + cw] += 1.0
            count = torch.from_numpy(full).to(dev)
        ops.resize_accumulate_chw(canvas, count, new_h, new_w, ph, pw, sh, sw, total, h, w, C,
                                  1.0 / len(self.scales))

    def _argmax(self, total, C, h, w):
        pred = torch.empty(h, w, dtype=torch.int64, device=self.device)
        ops.argmax_chw(total, pred, C, h, w)
        return pred

    @torch.no_grad()
    def predict(self, image_hwc, return_prob=False):
        """image_hwc: float32 [H,W,3] RGB in 0..255 (what SemData + ToTensor hand to test.py:188-190).

        With `shard=True` (opt-in: it is a collective) and torch.distributed initialised, every rank must pass the
        SAME image; the crops are sharded over the ranks and the [C,h,w] probability sums are combined by ONE reduce
        to rank 0 (`all_ranks=True`: all-reduce).  Ranks that do not receive the result return None.  The default
        (`shard=False`) is the reference's mode: each process handles whole images on its own (tool/test.py:88-93)."""
        dist, rank, world = self._dist()
        img = torch.as_tensor(image_hwc, dtype=torch.float32, device=self.device).contiguous()
        h, w, _ = img.shape
        C = self.classes
        plan = self.plan(h, w)
        mine = self.shard_units(plan, rank, world)
        total = self._new_total(C, h, w)
        for si in sorted(mine):
            self._accumulate_scale(img, h, w, plan[si], mine[si], total, world > 1)
        if world > 1:
            if self.all_ranks:
                dist.all_reduce(total, group=self.group)
            else:
                dist.reduce(total, dst=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                            group=self.group)
                if rank != 0:
                    return (None, None) if return_prob else None
        pred = self._argmax(total, C, h, w)
        return (pred, total) if return_prob else pred
