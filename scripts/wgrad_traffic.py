"""Per-shape HBM traffic of the weight-gradient kernel.  Two modes:
  python scripts/wgrad_traffic.py run            # under rocprofv3 --kernel-trace --pmc FETCH_SIZE (or WRITE_SIZE):
                                                 # every shape of the PSPNet-101 bs16 census, 2 launches each, fixed order
  python scripts/wgrad_traffic.py digest <fetch_counter_collection.csv> [<write_counter_collection.csv>]
Corrections as in profiles/*_pmc_per_kernel.json: counters are KB, FETCH_SIZE doubled on gfx950."""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # name, H, Ci, Co, k, stride, pad, dil, count(R101)
    ("l1 conv1 256->64 1x1 @119", 119, 256, 64, 1, 1, 0, 1, 2),
    ("l2 conv2 128->128 3x3 @60", 60, 128, 128, 3, 1, 1, 1, 3),
    ("l2 conv3 128->512 1x1 @60", 60, 128, 512, 1, 1, 0, 1, 4),
    ("l3 conv1 1024->256 1x1", 60, 1024, 256, 1, 1, 0, 1, 22),
    ("l3 conv2 256->256 3x3 d2", 60, 256, 256, 3, 1, 2, 2, 23),
    ("l3 conv3 256->1024 1x1", 60, 256, 1024, 1, 1, 0, 1, 23),
    ("l4 conv1 2048->512 1x1", 60, 2048, 512, 1, 1, 0, 1, 2),
    ("l4 conv2 512->512 3x3 d4", 60, 512, 512, 3, 1, 4, 4, 3),
    ("l4 conv3 512->2048 1x1", 60, 512, 2048, 1, 1, 0, 1, 3),
    ("l4 ds 1024->2048 1x1", 60, 1024, 2048, 1, 1, 0, 1, 1),
    ("cls.0 4096->512 3x3", 60, 4096, 512, 3, 1, 1, 1, 1),
    ("aux.0 1024->256 3x3", 60, 1024, 256, 3, 1, 1, 1, 1),
]
N = 16


def run():
    import torch
    from semseg_amd import ops
    dev = "cuda"
    scratch = torch.empty(96 * 1024 * 1024, device=dev)
    for name, H, Ci, Co, k, s, p, d, cnt in SHAPES:
        Ho = ops.conv_out(H, k, s, p, d)
        x = torch.randn(N, H, H, Ci, device=dev)
        dy = torch.randn(N, Ho, Ho, Co, device=dev)
        dw = torch.empty(Co, Ci, k, k, device=dev)
        for _ in range(2):
            ops.conv_wgrad(x, Ci, dy, Co, dw, scratch, N, H, H, Ci, Co, k, k, s, p, d)
        torch.cuda.synchronize()
        if os.environ.get("TIME"):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(3):
                e0.record()
                for _ in range(4):
                    ops.conv_wgrad(x, Ci, dy, Co, dw, scratch, N, H, H, Ci, Co, k, k, s, p, d)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 4 * 1e3)
            fl = 2.0 * N * Ho * Ho * Co * Ci * k * k
            print("%-28s %8.1f us %6.1f TF" % (name, min(ts), fl / min(ts) / 1e6), flush=True)
        del x, dy, dw


def per_dispatch(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if "conv_wgrad" not in n:
            continue
        did = int(r["Dispatch_Id"])
        rows[did] = rows.get(did, 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def digest(fetch_csv, write_csv=None):
    f = per_dispatch(fetch_csv, "FETCH_SIZE")
    w = per_dispatch(write_csv, "WRITE_SIZE") if write_csv else [0.0] * len(f)
    assert len(f) == 2 * len(SHAPES), (len(f), len(SHAPES))
    print("%-28s %9s %9s %9s %7s" % ("shape", "alg MB", "fetch MB", "write MB", "ratio"))
    tot_a = tot_t = 0.0
    for i, (name, H, Ci, Co, k, s, p, d, cnt) in enumerate(SHAPES):
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        alg = (N * H * H * Ci + N * Ho * Ho * Co + Co * Ci * k * k) * 4 / 1e6
        fe = f[2 * i + 1] * 1024 * 2 / 1e6          # second launch of the pair (first one also warms the code)
        wr = w[2 * i + 1] * 1024 / 1e6
        print("%-28s %9.1f %9.1f %9.1f %7.2f" % (name, alg, fe, wr, (fe + wr) / alg))
        tot_a += alg * cnt
        tot_t += (fe + wr) * cnt
    print("count-weighted: algorithmic %.1f MB, moved %.1f MB per launch-mix step; ratio %.2f" % (tot_a, tot_t, tot_t / tot_a))


def counters(path):
    """per-shape table of every counter in a counter_collection.csv (second launch of each pair)"""
    rows = {}
    for r in csv.DictReader(open(path)):
        if "conv_wgrad" not in r["Kernel_Name"]:
            continue
        rows.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
        rows[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(rows)
    assert len(ids) == 2 * len(SHAPES), len(ids)
    names = sorted({c for v in rows.values() for c in v})
    print("%-28s " % "shape" + " ".join("%16s" % n[-16:] for n in names))
    for i, sh in enumerate(SHAPES):
        v = rows[ids[2 * i + 1]]
        print("%-28s " % sh[0] + " ".join("%16.4g" % v.get(n, 0) for n in names))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "counters":
        counters(sys.argv[2])
    else:
        digest(*sys.argv[2:4])
