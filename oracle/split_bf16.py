"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

CPU restatement of the arithmetic of the split-bf16 experiment (kernels: csrc/gemm_bf16split.hip,
the SP instances of csrc/conv_igemm.hip): an fp32 value is cut into bf16 pieces by round-to-nearest-even at every level
(v_cvt_pk_bf16_f32), the remainders are formed in fp32 (exact), and a product is the fp32-accumulated sum of the leading
cross products of the pieces (bf16 x bf16 products are exact in fp32).  There is no reference file for this — the
reference computes in fp32 (model/pspnet.py, model/resnet.py: plain nn.Conv2d) — so what is pinned here is the error
model the experiment's claims rest on, against fp64."""
import numpy as np

SIX = ((1, 1), (0, 2), (2, 0), (0, 1), (1, 0), (0, 0))      # kernel order: small terms first, leading product last
THREE = ((0, 1), (1, 0), (0, 0))


def bf16_rne(x):
    """fp32 -> nearest-even bf16, returned as fp32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split(x, pieces):
    """x -> [p0, p1, ...] with p_k = bf16(x - p0 - ... - p_{k-1}); every subtraction is exact in fp32."""
    out, r = [], np.asarray(x, dtype=np.float32)
    for _ in range(pieces):
        p = bf16_rne(r)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def matmul_split(a, b, pieces):
    """a [M, K] @ b [K, N] from `pieces` bf16 pieces per operand: 3 cross products for two pieces, 6 for three;
    every partial matmul accumulates in fp32 like the matrix-core instruction's accumulator."""
    A, B = split(a, pieces), split(b, pieces)
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for i, j in (THREE if pieces == 2 else SIX):
        acc = acc + (A[i] @ B[j]).astype(np.float32)
    return acc
