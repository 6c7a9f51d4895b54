"""CPU: the error model of the split-bf16 experiment, pinned with oracle/split_bf16.py."""
import numpy as np

from oracle import split_bf16 as sb


def _rms(x, ref):
    return float(np.sqrt(np.mean((x.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))


def test_bf16_rounding_matches_torch():
    import torch
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    x[:4] = [0.0, -0.0, 1.0, np.float32(1.0) + np.float32(2.0 ** -8)]      # the last one is a tie: rounds to even
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(sb.bf16_rne(x).view(np.uint32), ref.view(np.uint32))


def test_three_pieces_reproduce_fp32_and_two_carry_16_bits():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(500000) * np.exp(rng.uniform(-10, 10, 500000))).astype(np.float32)
    h, m, l = sb.split(x, 3)
    # 8 + 8 + 8 mantissa bits with round-to-nearest pieces: the sum is the fp32 value itself
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
    h2, l2 = sb.split(x, 2)
    err = np.abs(x.astype(np.float64) - h2 - l2) / np.abs(x)
    assert err.max() <= 2.0 ** -17 * 1.0001 and err.max() > 2.0 ** -19


def test_six_products_are_fp32_grade_three_are_not():
    """GEMM of the cls.0 data-gradient shape (K = 512) and a long one (K = 4096): post-ReLU activations x Kaiming
    weights.  Six cross products: rms error within 1.3x of a plain fp32 matmul's; three: ~10x (what the GPU probe
    measures too: 3.4e-7 / 4.4e-6 against 4.1e-7 for the fp32 kernel, profiles/r03_split_bf16_probe.txt)."""
    rng = np.random.default_rng(2)
    for K in (512, 4096):
        a = np.maximum(rng.standard_normal((192, K)), 0).astype(np.float32)
        b = (rng.standard_normal((K, 160)) * (2.0 / K) ** 0.5).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        e32 = _rms(a @ b, ref)
        e6 = _rms(sb.matmul_split(a, b, 3), ref)
        e3 = _rms(sb.matmul_split(a, b, 2), ref)
        assert e6 <= 1.3 * e32 + 2e-8, (K, e6, e32)
        assert 4 * e32 < e3 < 1e-5, (K, e3, e32)


def test_dropped_products_are_below_fp32_resolution():
    """What the six-product form leaves out (m*l, l*m, l*l) relative to the exact product of two fp32 values."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal(200000).astype(np.float32)
    b = rng.standard_normal(200000).astype(np.float32)
    A, B = sb.split(a, 3), sb.split(b, 3)
    six = sum(A[i].astype(np.float64) * B[j] for i, j in sb.SIX)
    exact = a.astype(np.float64) * b
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -23
