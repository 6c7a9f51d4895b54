"""The two bf16x3 row-GEMM kernels on the batched GEMM shapes of the Winograd path (16 GEMMs [T x K] x [K x N] per launch): the
256 x 128 kernel of csrc/gemm_bf16split.hip against the SP instances of conv_igemm_kernel (128 x 128), per shape, interleaved.
python scripts/wino_gemm_ab.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda"
shapes = [("layer2 conv2 128->128 d1", 60, 1, 128, 128), ("layer3 conv2 256->256 d2", 60, 2, 256, 256),
          ("layer4 conv2 512->512 d4", 60, 4, 512, 512), ("aux.0 1024->256", 60, 1, 1024, 256),
          ("cls.0 fwd 4096->512", 60, 1, 4096, 512), ("cls.0 dgrad 512->4096", 60, 1, 512, 4096)]


def timeit(fn, it=6):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for name, H, d, K, N in shapes:
    T = ops.wino_tiles(B, H, H, d)
    V = torch.randn(16 * T * K, device=dev)
    rows_pad = ops.roundup(N, 128)
    U = torch.randn(16 * rows_pad * K, device=dev) / K ** 0.5
    M1 = torch.empty(16 * T * N, device=dev)
    M2 = torch.empty(16 * T * N, device=dev)
    f_std = lambda: ops.gemm_rows_batched_bf16split(V, K, T * K, U, rows_pad * K, M1, N, T * N, T, K, N, 16, nsplit=3)
    f_ig = lambda: ops.gemm_rows_batched(V, K, T * K, U, rows_pad * K, M2, N, T * N, T, K, N, 16, arith=ops.ARITH_BF16X3)
    f_f32 = lambda: ops.gemm_rows_batched(V, K, T * K, U, rows_pad * K, M2, N, T * N, T, K, N, 16)
    res = {"std": [], "igemm": [], "f32": []}
    for _ in range(3):
        res["std"].append(timeit(f_std)); res["igemm"].append(timeit(f_ig)); res["f32"].append(timeit(f_f32))
    fl = 2.0 * 16 * T * K * N
    f_ig(); f_std(); torch.cuda.synchronize()
    diff = float((M1 - M2).abs().max() / M2.abs().max())
    print("%-28s T %6d  256x128 kernel %8.1f us (%6.1f TF)   igemm SP %8.1f us (%6.1f TF)   fp32 %8.1f us (%6.1f TF)   max rel diff %.1e"
          % (name, T, min(res["std"]), fl / min(res["std"]) / 1e6, min(res["igemm"]), fl / min(res["igemm"]) / 1e6,
             min(res["f32"]), fl / min(res["f32"]) / 1e6, diff), flush=True)
