"""TEST INFRASTRUCTURE — numpy/ctypes front-end of oracle/psamask_oracle.c (restatement of
/root/reference/lib/psa/src/cpu/psamask.cpp:11-133 and the Python glue
/root/reference/lib/psa/functions/psamask.py:8-36)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(HERE, "libpsamask_oracle.so")
_dll = None


def build():
    src = os.path.join(HERE, "psamask_oracle.c")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _LIB, src])
    return _LIB


def _lib():
    global _dll
    if _dll is None:
        build()
        _dll = ctypes.CDLL(_LIB)
        for f in (_dll.oracle_psamask_forward, _dll.oracle_psamask_backward):
            f.restype = None
            f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7
    return _dll


def psa_mask_forward(inp, psa_type=0, mask_H=None, mask_W=None):
    """inp: float32 [N, mH*mW, H, W] -> [N, H*W, H, W] (functions/psamask.py:8-24)."""
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    n, c, H, W = inp.shape
    if mask_H is None and mask_W is None:
        mask_H, mask_W = 2 * H - 1, 2 * W - 1
    assert mask_H % 2 == 1 and mask_W % 2 == 1 and c == mask_H * mask_W
    out = np.zeros((n, H * W, H, W), dtype=np.float32)
    _lib().oracle_psamask_forward(psa_type, inp.ctypes.data, out.ctypes.data, n, H, W, mask_H, mask_W,
                                  (mask_H - 1) // 2, (mask_W - 1) // 2)
    return out


def psa_mask_backward(grad_out, psa_type, mask_H, mask_W):
    """grad_out: float32 [N, H*W, H, W] -> [N, mH*mW, H, W] (functions/psamask.py:28-36)."""
    grad_out = np.ascontiguousarray(grad_out, dtype=np.float32)
    n, hw, H, W = grad_out.shape
    gin = np.zeros((n, mask_H * mask_W, H, W), dtype=np.float32)
    _lib().oracle_psamask_backward(psa_type, grad_out.ctypes.data, gin.ctypes.data, n, H, W, mask_H,
                                   mask_W, (mask_H - 1) // 2, (mask_W - 1) // 2)
    return gin
