"""The Winograd F(2x2, 3x3) algorithm of csrc/winograd.hip restated on the CPU (oracle/winograd.py: same tile numbering,
dilation phases, transform matrices, zero handling) against torch's direct convolution and its gradients in float64:
pins the algorithm itself where no GPU exists (the HIP kernels are checked against fp64 F.conv2d in test_ops_gpu.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import winograd as wg


@pytest.mark.parametrize("N,H,W,Ci,Co,d", [(2, 8, 8, 3, 4, 1), (1, 9, 7, 2, 3, 1), (1, 11, 10, 3, 2, 2), (1, 13, 13, 2, 2, 4),
                                           (2, 5, 6, 4, 1, 3)])
def test_winograd_restated_equals_direct_conv(N, H, W, Ci, Co, d):
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + d)
    x = torch.randn(N, Ci, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Co, Ci, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(N, Co, H, W, generator=g, dtype=torch.float64)
    y = F.conv2d(x, w, None, 1, d, d)
    y.backward(dy)
    xn, wn, dyn = x.detach().numpy(), w.detach().numpy(), dy.numpy()
    assert np.allclose(wg.conv_forward(xn, wn, d), y.detach().numpy(), atol=1e-12)
    assert np.allclose(wg.conv_dgrad(dyn, wn, d), x.grad.numpy(), atol=1e-12)
    assert np.allclose(wg.conv_wgrad(xn, dyn, d), w.grad.numpy(), atol=1e-11)


def test_tile_geometry_covers_every_pixel_once():
    for (N, H, W, d) in [(2, 60, 60, 1), (1, 60, 60, 2), (1, 60, 60, 4), (1, 90, 90, 4), (1, 15, 13, 2), (3, 5, 5, 3)]:
        _, _, T = wg.geometry(N, H, W, d)
        seen = np.zeros((N, H, W), dtype=np.int32)
        for t in range(T):
            n, y0, x0 = wg.tile_origin(t, N, H, W, d)
            for a in range(2):
                for b in range(2):
                    yy, xx = y0 + a * d, x0 + b * d
                    if yy < H and xx < W:
                        seen[n, yy, xx] += 1
        assert (seen == 1).all(), (N, H, W, d)


def test_c_abi_tile_count_matches_the_restatement():
    """semseg_wino_tiles is host code of the C-ABI library (no kernel launch): same tile count as the restatement."""
    from semseg_amd._lib import lib
    for (N, H, W, d) in [(16, 60, 60, 1), (16, 60, 60, 2), (16, 60, 60, 4), (2, 90, 90, 2), (2, 90, 90, 4), (1, 15, 13, 2)]:
        assert lib.semseg_wino_tiles(N, H, W, d) == wg.geometry(N, H, W, d)[2]
    assert lib.semseg_wino_tiles(0, 60, 60, 1) == -1
