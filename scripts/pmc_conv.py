import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops
dev = "cuda"; N = 16
scratch = torch.empty(64 * 1024 * 1024, device=dev)
for name, H, Ci, Co, k, s, p, d in [("l4conv2", 60, 512, 512, 3, 1, 4, 4), ("l3conv2", 60, 256, 256, 3, 1, 2, 2), ("l3conv1", 60, 1024, 256, 1, 1, 0, 1)]:
    pk = ops.PackedConv(Co, Ci, k, k, dev); w = torch.randn(Co, Ci, k, k, device=dev) * 0.05; pk.pack(w)
    x = torch.randn(N, H, H, Ci, device=dev); y = torch.zeros(N, H, H, Co, device=dev)
    dy = torch.randn(N, H, H, Co, device=dev); dx = torch.empty(N, H, H, Ci, device=dev); dw = torch.empty(Co, Ci, k, k, device=dev)
    stats = torch.zeros(2 * Co, dtype=torch.float64, device=dev)
    for _ in range(3):
        ops.conv_fwd(x, Ci, pk, y, Co, N, H, H, s, p, d, stats=stats)
        ops.conv_dgrad(dy, Co, pk, dx, Ci, N, H, H, s, p, d)
        ops.conv_wgrad(x, Ci, dy, Co, dw, scratch, N, H, H, Ci, Co, k, k, s, p, d)
torch.cuda.synchronize()
