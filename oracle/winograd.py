"""CPU restatement (test infrastructure only) of the Winograd F(2x2, 3x3) algorithm that semseg_amd/csrc/winograd.hip
implements for the reference's stride-1 3x3 convolutions (nn.Conv2d(k=3, stride=1, padding=dilation, dilation=dilation):
model/resnet.py:63-69 after model/pspnet.py:49-58; head convs model/pspnet.py:65,73).  numpy, float64 by default; the
same tile numbering, phase decomposition for dilation, transform matrices and zero handling as the kernels, so that the
algorithm itself (not only its HIP implementation) is pinned against F.conv2d on the CPU.

    Y = A^T [ (G g G^T) .* (B^T d B) ] A          Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks"
"""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
G = np.array([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)


def geometry(N, H, W, d):
    """(th, tw, T): tiles per phase row / column and in total, as make_geo() in winograd.hip."""
    th = ((H + d - 1) // d + 1) // 2
    tw = ((W + d - 1) // d + 1) // 2
    return th, tw, N * d * d * th * tw


def tile_origin(t, N, H, W, d):
    """tile index -> (n, y0, x0): image row / column of the tile's first output pixel (decode_tile in winograd.hip)."""
    th, tw, _ = geometry(N, H, W, d)
    tx = t % tw; t //= tw
    ty = t % th; t //= th
    rx = t % d; t //= d
    ry = t % d
    n = t // d
    return n, d * 2 * ty + ry, d * 2 * tx + rx


def input_transform(x, d):
    """x [N, C, H, W] -> V [16, T, C]  (patch rows / columns outside the image are zero)."""
    N, C, H, W = x.shape
    _, _, T = geometry(N, H, W, d)
    V = np.zeros((16, T, C), dtype=x.dtype)
    for t in range(T):
        n, y0, x0 = tile_origin(t, N, H, W, d)
        patch = np.zeros((4, 4, C), dtype=x.dtype)
        for i in range(4):
            for j in range(4):
                y, xx = y0 + (i - 1) * d, x0 + (j - 1) * d
                if 0 <= y < H and 0 <= xx < W:
                    patch[i, j] = x[n, :, y, xx]
        v = np.einsum("ik,klc,jl->ijc", BT, patch, BT)
        V[:, t, :] = v.reshape(16, C)
    return V


def filter_transform(w, flip=False):
    """w [Co, Ci, 3, 3] -> U [16, Co, Ci]; flip: the data gradient's filter U [16, Ci, Co] (taps rotated 180 degrees)."""
    if flip:
        w = np.flip(w, (2, 3)).transpose(1, 0, 2, 3)
    u = np.einsum("ik,ockl,jl->ijoc", G, w, G)
    return u.reshape(16, w.shape[0], w.shape[1])


def output_transform(M, N, H, W, d):
    """M [16, T, Co] -> y [N, Co, H, W]."""
    Co = M.shape[2]
    _, _, T = geometry(N, H, W, d)
    y = np.zeros((N, Co, H, W), dtype=M.dtype)
    for t in range(T):
        n, y0, x0 = tile_origin(t, N, H, W, d)
        o = np.einsum("ik,klc,jl->ijc", AT, M[:, t, :].reshape(4, 4, Co), AT)
        for a in range(2):
            for b in range(2):
                yy, xx = y0 + a * d, x0 + b * d
                if yy < H and xx < W:
                    y[n, :, yy, xx] = o[a, b]
    return y


def conv_forward(x, w, d):
    V, U = input_transform(x, d), filter_transform(w)
    M = np.einsum("etc,eoc->eto", V, U)
    return output_transform(M, x.shape[0], x.shape[2], x.shape[3], d)


def conv_dgrad(dy, w, d):
    V, U = input_transform(dy, d), filter_transform(w, flip=True)
    M = np.einsum("etk,eck->etc", V, U)
    return output_transform(M, dy.shape[0], dy.shape[2], dy.shape[3], d)


def conv_wgrad(x, dy, d):
    """dw [Co, Ci, 3, 3] = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G."""
    N, Ci, H, W = x.shape
    Co = dy.shape[1]
    _, _, T = geometry(N, H, W, d)
    V = input_transform(x, d)
    Yh = np.zeros((16, T, Co), dtype=dy.dtype)
    for t in range(T):
        n, y0, x0 = tile_origin(t, N, H, W, d)
        q = np.zeros((2, 2, Co), dtype=dy.dtype)
        for a in range(2):
            for b in range(2):
                yy, xx = y0 + a * d, x0 + b * d
                if yy < H and xx < W:
                    q[a, b] = dy[n, :, yy, xx]
        Yh[:, t, :] = np.einsum("ki,klc,lj->ijc", AT, q, AT).reshape(16, Co)      # A q A^T with A = AT^T
    dU = np.einsum("eto,etc->eoc", Yh, V).reshape(4, 4, Co, Ci)
    return np.einsum("ki,kloc,lj->ocij", G, dU, G)
