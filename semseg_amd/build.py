"""Build libsemseg_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (see include/semseg_hip.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsemseg_hip.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad.hip", "stem.hip", "bn.hip", "pool_interp.hip", "ce_head.hip", "psamask.hip",
           "psa_ops.hip", "infer.hip", "optim.hip", "augment.hip", "winograd.hip", "gemm_bf16split.hip", "xchg.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-shared"]


OBJ_DIR = os.path.join(CSRC, "build")          # git-ignored; objects are an incremental-build cache only


def _deps(src):
    return [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "conv_common.h"),
            os.path.join(HERE, "..", "include", "semseg_hip.h")]


def _linked_sources():
    """Names of the sources the library on disk was linked from (written next to it at link time)."""
    try:
        return open(LIB + ".sources").read().split()
    except OSError:
        return []


def needs_build():
    if not os.path.exists(LIB) or _linked_sources() != [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]:
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for s in SOURCES for d in _deps(s) if os.path.exists(d))


def build(force=False, verbose=False, defines=(), out=None):
    """One hipcc -c per source, in parallel, then one link.  Objects are cached per (source, defines) under csrc/build/ so
    that touching one kernel file rebuilds that file only (conv_igemm.hip and conv_wgrad.hip take about a minute each).  `defines` / `out`
    build a VARIANT library for A/B runs (scripts/; selected with SEMSEG_HIP_LIB), never the product."""
    out = out or LIB
    if not force and not defines and out == LIB and not needs_build():
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    tag = hashlib.sha1(" ".join(sorted(defines)).encode()).hexdigest()[:8] if defines else "base"
    cflags = [f for f in FLAGS if f != "-shared"] + ["-D" + d for d in defines]
    jobs = []
    for src in SOURCES:
        if not os.path.exists(os.path.join(CSRC, src)):
            continue
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (src, tag))
        stale = force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps(src))
        jobs.append((src, obj, stale))

    def compile_one(job):
        src, obj, stale = job
        if stale:
            import time
            t0 = time.time()
            cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            os.replace(obj + ".tmp", obj)
            # the object is as old as the moment its compile STARTED: a source edited while hipcc was running (its host and
            # device passes read the file separately) is newer than the object and gets rebuilt next time
            os.utime(obj, (t0, t0))
        return obj
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        objs = list(ex.map(compile_one, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(out + ".sources", "w") as f:
        f.write("\n".join(j[0] for j in jobs) + "\n")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
