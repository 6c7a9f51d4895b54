"""Per-workgroup phase timestamps of the forward kernel (experiment build -DEXP_TRACE, gpurun_variants/lib_trace.so):
start, K loop entered, first K-step done, K loop done, epilogue done (100 MHz wall clock) + XCC / HW_ID of wave 0."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SEMSEG_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "lib_trace.so")
import numpy as np, torch
from semseg_amd import ops
from semseg_amd._lib import lib
dll = lib.load()._dll
dll.semseg_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 16
LAB = ["prologue (start -> loads issued)", "K-step 1", "K-step 2", "K-steps 3..", "epi col 0: acc -> LDS + fp64 stats", "epi col 0: LDS -> 8 stores",
       "epi col 1: acc -> LDS + fp64 stats", "epi col 1: LDS -> 8 stores", "epi: barriers + red + atomics"]
for name, H, Ci, Co in (("l3 conv3 256->1024", 60, 256, 1024), ("l4 conv3 512->2048", 60, 512, 2048)):
    pk = ops.PackedConv(Co, Ci, 1, 1, "cuda"); pk.pack(torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05)
    x = torch.randn(N, H, H, Ci, device="cuda"); y = torch.zeros(N, H, H, Co, device="cuda")
    st = torch.zeros(2 * Co * ops.NSLOT, dtype=torch.float64, device="cuda")
    scr = torch.empty(64 * 1024 * 1024, device="cuda")
    for _ in range(2):
        ops.conv_fwd(x, Ci, pk, y, Co, N, H, H, 1, 0, 1, stats=st, nslot=ops.NSLOT, scratch=scr)
    torch.cuda.synchronize()
    buf = np.zeros(16384 * 12, dtype=np.uint64)
    assert dll.semseg_debug_read_trace(buf.ctypes.data, buf.size) == 0      # drop the warm-up launches' stamps
    ops.conv_fwd(x, Ci, pk, y, Co, N, H, H, 1, 0, 1, stats=st, nslot=ops.NSLOT, scratch=scr)
    torch.cuda.synchronize()
    assert dll.semseg_debug_read_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(-1, 12)
    t = t[t[:, 0] > 0]
    n = len(t)
    t0 = t[:, 0].min()
    ts = (t[:, :10].astype(np.int64) - int(t0)) / 100.0     # us
    print("== %s: %d workgroups, kernel span %.1f us, lifetime median %.1f us" % (name, n, ts[:, 9].max(), np.median(ts[:, 9] - ts[:, 0])))
    d = np.diff(ts, axis=1)
    for i, lab in enumerate(LAB):
        print("  %-48s median %6.2f  p10 %6.2f  p90 %6.2f us" % (lab, np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
