"""Native operator package of lib/psa — same surface as the reference's lib/psa/src/__init__.py:9-18: attribute
`gpu` is the pybind module `psamask_gpu`, built with torch.utils.cpp_extension into lib/psa/src/gpu/ and exporting
`psamask_forward` / `psamask_backward` with the `(int, at::Tensor const&, at::Tensor&, 7 x int)` signature of
lib/psa/src/gpu/operator.h:3-4.  Here the module is a thin at::Tensor front-end over the C ABI of libsemseg_hip.so
(include/semseg_hip.h), so the reference's lib/psa/functions/psamask.py runs on the gfx950 kernels unmodified.

There is no `cpu` module: this build has no CPU path (touching `src.cpu` raises).  The extension is compiled in-tree
(`python -m lib.psa.src` or __graft_entry__.build()); at import an up-to-date psamask_gpu.so is loaded directly,
otherwise torch.utils.cpp_extension.load builds it (needs the ROCm headers, ~1 min).
"""
import importlib.util
import os
import sys

import torch  # noqa: F401  (libtorch / libamdhip64 must be resident before the extension resolves its symbols)

cwd = os.path.dirname(os.path.realpath(__file__))
gpu_path = os.path.join(cwd, "gpu")
_ROOT = os.path.normpath(os.path.join(cwd, "..", "..", ".."))
_INCLUDE = os.path.join(_ROOT, "include")
_LIBDIR = os.path.join(_ROOT, "semseg_amd", "csrc")
_SOURCES = [os.path.join(gpu_path, "operator.cpp")]
_DEPS = _SOURCES + [os.path.join(gpu_path, "operator.h"), os.path.join(_INCLUDE, "semseg_hip.h")]
_SO = os.path.join(gpu_path, "psamask_gpu.so")


def _fresh():
    return os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in _DEPS)


def build(verbose=False):
    """torch.utils.cpp_extension build of psamask_gpu.so (one C++ source; links libsemseg_hip.so by rpath)."""
    from torch.utils.cpp_extension import ROCM_HOME, load
    if not os.path.exists(os.path.join(_LIBDIR, "libsemseg_hip.so")):
        raise RuntimeError("build libsemseg_hip.so first: python -m semseg_amd.build")
    rocm = ROCM_HOME or "/opt/rocm"
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    return load("psamask_gpu", _SOURCES, build_directory=gpu_path, verbose=verbose, with_cuda=False,
                extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
                extra_include_paths=[_INCLUDE, os.path.join(rocm, "include")],
                extra_ldflags=["-L" + _LIBDIR, "-lsemseg_hip", "-Wl,-rpath," + _LIBDIR,
                               "-L" + tlib, "-lc10_hip", "-ltorch_hip", "-L" + os.path.join(rocm, "lib"),
                               "-lamdhip64"])


def _load():
    if _fresh():
        spec = importlib.util.spec_from_file_location("psamask_gpu", _SO)
        mod = importlib.util.module_from_spec(spec)
        # libsemseg_hip.so is found through the rpath recorded at build time; fall back to an explicit preload
        # when the tree was moved after the build
        import ctypes
        try:
            spec.loader.exec_module(mod)
        except ImportError:
            ctypes.CDLL(os.path.join(_LIBDIR, "libsemseg_hip.so"), mode=ctypes.RTLD_GLOBAL)
            spec.loader.exec_module(mod)
        sys.modules.setdefault("psamask_gpu", mod)
        return mod
    return build()


gpu = _load()


def __getattr__(name):
    if name == "cpu":
        raise RuntimeError("lib.psa.src has no CPU module in the MI355X build: move the tensors to cuda "
                           "(semseg_amd has no CPU path)")
    raise AttributeError(name)


if __name__ == "__main__":
    print(build(verbose=True).__file__)
