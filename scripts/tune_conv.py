"""Build -D variants of the conv kernels and time them on the key PSPNet101 shapes.
usage: python scripts/tune_conv.py build   (CPU container)  |  python scripts/tune_conv.py run  (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semseg_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_variants")
VARIANTS = {
    "base": [],
    "occ4": ["-DEXP_OCC4=1"],
}
SRCS = ["conv_igemm.hip", "stem.hip", "bn.hip", "pool_interp.hip", "ce_head.hip", "psamask.hip", "psa_ops.hip", "infer.hip", "optim.hip", "augment.hip"]
if sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for tag, flags in VARIANTS.items():
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared"] + flags + \
              [os.path.join(CSRC, s) for s in SRCS] + ["-o", os.path.join(OUT, "lib_%s.so" % tag)]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0
else:
    for tag in VARIANTS:
        env = dict(os.environ, SEMSEG_HIP_LIB=os.path.join(OUT, "lib_%s.so" % tag))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "conv_bench.py")], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.split("\n") if any(k in l for k in ("l1 conv", "l3 conv", "l4 conv", "cls.0", "aux.0", "weighted"))]
        print("==", tag); print("\n".join(lines)); sys.stdout.flush()
