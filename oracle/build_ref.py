"""TEST INFRASTRUCTURE — compile the reference's own CPU psamask operator from the sources where
they lie (/root/reference/lib/psa/src/cpu/{operator.cpp,psamask.cpp}) into oracle/_ref/ (git-ignored,
but shipped to the GPU box by gpurun).  Nothing is copied into the repo.  Needs torch's C++ headers
(the reference binds at::Tensor), so the recipe is torch.utils.cpp_extension.load — the reference's
own build recipe (lib/psa/src/__init__.py:9-12) pointed at a writable build directory.

Usage: python oracle/build_ref.py   (no-op with a notice when /root/reference is absent)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SEMSEG_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def build(verbose=False):
    src = os.path.join(REF, "lib", "psa", "src", "cpu")
    if not os.path.isdir(src):
        print("oracle/_ref: %s not present, using prebuilt files if any" % src, file=sys.stderr)
        return None
    so = os.path.join(OUT, "psamask_cpu_ref.so")
    srcs = [os.path.join(src, "operator.cpp"), os.path.join(src, "psamask.cpp")]
    if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    load("psamask_cpu_ref", srcs, build_directory=OUT, verbose=verbose)
    return so


def load_ref():
    """Import the prebuilt reference extension (works without /root/reference)."""
    so = os.path.join(OUT, "psamask_cpu_ref.so")
    if not os.path.exists(so):
        return None
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location("psamask_cpu_ref", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose=True))
