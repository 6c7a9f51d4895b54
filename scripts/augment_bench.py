"""Device-side input pipeline (semseg_amd/transform.py) throughput next to the CPU restatement of the reference's
cv2 chain (oracle/transform.py, one core) on the same samples.  Usage: python scripts/augment_bench.py [batch] [reps]
Prints one JSON line per dataset shape."""
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import transform_cases as tc                 # noqa: E402
from semseg_amd import transform as T        # noqa: E402
from oracle import transform as otf          # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for name, H, W, crop in (("ade20k 512x683 -> 473", 512, 683, 473), ("cityscapes 1024x2048 -> 713", 1024, 2048, 713)):
        rng = np.random.default_rng(0)
        imgs = [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(B)]
        labs = [rng.integers(0, 150, size=(H, W), dtype=np.uint8) for _ in range(B)]
        ops = tc.train_chain((crop, crop))
        chain = tc.build_chain(T, ops)
        dimgs = [torch.from_numpy(i).cuda() for i in imgs]
        dlabs = [torch.from_numpy(l).cuda() for l in labs]
        random.seed(0)
        for _ in range(3):
            chain.batch(imgs, labs)
        torch.cuda.synchronize()
        out = {"workload": name, "batch": B}
        for key, src in (("host_u8_sources", (imgs, labs)), ("hbm_u8_sources", (dimgs, dlabs))):
            random.seed(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                x, y = chain.batch(*src)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            out[key] = {"ms_per_batch": round(dt * 1e3, 3), "images_per_s": round(B / dt, 1)}
        # kernels only: replay one planned batch with events
        random.seed(2)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import semseg_amd.ops as ops_mod
        spans = []
        orig = ops_mod.augment_round

        def timed(ops_dev, n, mp):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            orig(ops_dev, n, mp)
            b.record()
            spans.append((a, b))
        ops_mod.augment_round = timed
        T.ops.augment_round = timed
        for _ in range(reps):
            chain.batch(dimgs, dlabs)
        torch.cuda.synchronize()
        ops_mod.augment_round = orig
        T.ops.augment_round = orig
        kern_ms = sum(a.elapsed_time(b) for a, b in spans) / reps
        in_bytes = B * H * W * 4
        out_bytes = B * crop * crop * (12 + 8)
        out["kernels_ms_per_batch"] = round(kern_ms, 4)
        out["launches_per_batch"] = len(spans) // reps
        out["algorithmic_GB_per_s"] = round((in_bytes + out_bytes) / kern_ms / 1e6, 1)
        out["algorithmic_bytes"] = "decoded uint8 image+label read once + float CHW/int64 batch written once"
        # CPU restatement of the reference chain, one core, bounded sample
        torch.set_num_threads(1)
        random.seed(1)
        n = 0
        t0 = time.perf_counter()
        while n < 4 or time.perf_counter() - t0 < 5.0:
            otf.run(ops, np.float32(imgs[n % B]), labs[n % B].copy())
            n += 1
        dt = (time.perf_counter() - t0) / n
        out["cpu_port_1core"] = {"ms_per_image": round(dt * 1e3, 1), "images_per_s": round(1 / dt, 2), "samples": n,
                                 "kind": "numpy restatement of the cv2 chain (oracle/), not cv2 itself"}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
