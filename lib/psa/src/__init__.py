"""Native operator package of lib/psa — same surface as the reference's lib/psa/src/__init__.py:9-18: attribute
`gpu` is the pybind module `psamask_gpu`, built with torch.utils.cpp_extension into lib/psa/src/gpu/ and exporting
`psamask_forward` / `psamask_backward` with the `(int, at::Tensor const&, at::Tensor&, 7 x int)` signature of
lib/psa/src/gpu/operator.h:3-4.  Here the module is a thin at::Tensor front-end over the C ABI of libsemseg_hip.so
(include/semseg_hip.h), so the reference's lib/psa/functions/psamask.py runs on the gfx950 kernels unmodified.

There is no `cpu` module: this build has no CPU path (touching `src.cpu` raises).  The extension is compiled in-tree
(`python -m lib.psa.src` or __graft_entry__.build()).  `src.gpu` is resolved on FIRST USE: an up-to-date psamask_gpu.so
(content hash of its sources) is loaded directly, otherwise torch.utils.cpp_extension.load builds it (needs the ROCm
headers, ~1 min); if that fails the same two entry points are served through the ctypes binding of libsemseg_hip.so.
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


_DIGEST = os.path.join(gpu_path, ".psamask_gpu.sha256")


def _digest():
    """Content hash of everything the extension is built from (sources, headers, the torch build it links)."""
    import hashlib
    h = hashlib.sha256(torch.__version__.encode())
    for d in _DEPS:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _fresh():
    """An existing psamask_gpu.so is reused when it was built from the current sources (content hash, not mtimes: a
    checkout touches every file)."""
    if not os.path.exists(_SO):
        return False
    try:
        with open(_DIGEST) as f:
            return f.read().strip() == _digest()
    except OSError:
        return False


def build(verbose=False):
    """torch.utils.cpp_extension build of psamask_gpu.so (one C++ source; links libsemseg_hip.so by rpath)."""
    from torch.utils.cpp_extension import ROCM_HOME, load
    if not os.path.exists(os.path.join(_LIBDIR, "libsemseg_hip.so")):
        raise RuntimeError("build libsemseg_hip.so first: python -m semseg_amd.build")
    rocm = ROCM_HOME or "/opt/rocm"
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    mod = load("psamask_gpu", _SOURCES, build_directory=gpu_path, verbose=verbose, with_cuda=False,
                extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1"],
                extra_include_paths=[_INCLUDE, os.path.join(rocm, "include")],
                extra_ldflags=["-L" + _LIBDIR, "-lsemseg_hip", "-Wl,-rpath," + _LIBDIR,
                               "-L" + tlib, "-lc10_hip", "-ltorch_hip", "-L" + os.path.join(rocm, "lib"),
                               "-lamdhip64"])
    with open(_DIGEST, "w") as f:
        f.write(_digest() + "\n")
    return mod


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


class _CtypesFallback(object):
    """Same two entry points through the C ABI of libsemseg_hip.so directly (semseg_amd.ops, ctypes) — used only when
    the pybind extension cannot be built on this machine (no ROCm headers / ninja, another torch library layout), so
    that importing lib.psa / model.psanet never depends on a C++ toolchain.  Still the gfx950 kernels, still no CPU path."""
    __name__ = "psamask_gpu"
    __file__ = "<ctypes fallback over libsemseg_hip.so>"

    @staticmethod
    def _check(src, dst, what):
        if not (src.is_cuda and dst.is_cuda):
            raise RuntimeError(what + ": tensors must live on the MI355X (no CPU path in psamask_gpu)")
        if src.dtype != torch.float32 or dst.dtype != torch.float32:
            raise RuntimeError(what + ": float32 only")
        if not (src.is_contiguous() and dst.is_contiguous()):
            raise RuntimeError(what + ": dense NCHW tensors expected")

    def psamask_forward(self, psa_type, input, output, num_, fH, fW, mH, mW, hH, hW):
        from semseg_amd import ops
        self._check(input, output, "psamask_forward")
        if tuple(input.shape) != (num_, mH * mW, fH, fW) or tuple(output.shape) != (num_, fH * fW, fH, fW):
            raise RuntimeError("psamask_forward: tensor shapes do not match the geometry arguments")
        ops.psamask_forward(psa_type, input, output, num_, fH, fW, mH, mW, hH, hW)

    def psamask_backward(self, psa_type, grad_output, grad_input, num_, fH, fW, mH, mW, hH, hW):
        from semseg_amd import ops
        self._check(grad_output, grad_input, "psamask_backward")
        if tuple(grad_output.shape) != (num_, fH * fW, fH, fW) or tuple(grad_input.shape) != (num_, mH * mW, fH, fW):
            raise RuntimeError("psamask_backward: tensor shapes do not match the geometry arguments")
        ops.psamask_backward(psa_type, grad_output, grad_input, num_, fH, fW, mH, mW, hH, hW)


_gpu = None


def __getattr__(name):
    # `gpu` is resolved on first use (PEP 562): importing lib.psa / model.psanet costs nothing, the PSANet engine path never
    # touches this module, and a machine without a C++ toolchain still gets the operator through the ctypes fallback
    global _gpu
    if name == "gpu":
        if _gpu is None:
            try:
                _gpu = _load()
            except Exception as e:   # build / link failure: keep the operator usable
                import warnings
                warnings.warn("lib.psa.src: could not build / load the psamask_gpu extension (%s: %s); using the ctypes "
                              "binding of libsemseg_hip.so instead" % (type(e).__name__, e))
                _gpu = _CtypesFallback()
        return _gpu
    if name == "cpu":
        raise RuntimeError("lib.psa.src has no CPU module in the MI355X build: move the tensors to cuda "
                           "(semseg_amd has no CPU path)")
    raise AttributeError(name)


if __name__ == "__main__":
    print(build(verbose=True).__file__)
