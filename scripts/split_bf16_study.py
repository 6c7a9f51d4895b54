"""CPU-only numerical study for the round-3 candidate "split-bf16" (VERDICT r1 item 9): how close does a GEMM get
to fp32 when each fp32 operand is split into 3 bf16 pieces and the 6 leading cross products are accumulated in fp32
(what the bf16 MFMA does: exact bf16 x bf16 products, fp32 accumulation)?  No GPU involved; torch CPU only.
python scripts/split_bf16_study.py"""
import torch

torch.manual_seed(0)
torch.set_num_threads(8)


def split3(x):
    p1 = x.to(torch.bfloat16)
    r = x - p1.float()
    p2 = r.to(torch.bfloat16)
    r = r - p2.float()
    p3 = r.to(torch.bfloat16)
    return [p1.float(), p2.float(), p3.float()]


def gemm_split(a, b, terms):
    A, B = split3(a), split3(b)
    out = torch.zeros(a.shape[0], b.shape[1], dtype=torch.float32)
    # smallest terms first, as a kernel that cares about rounding would order them
    for i, j in sorted(terms, key=lambda t: -(t[0] + t[1])):
        out += A[i] @ B[j]
    return out


def rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max()), float(((x.double() - ref) ** 2).mean().sqrt() /
                                                                          (ref ** 2).mean().sqrt())


SIX = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
THREE = [(0, 0), (0, 1), (1, 0)]
NINE = [(i, j) for i in range(3) for j in range(3)]
print("%-28s %10s | %-21s | %-21s | %-21s | %-21s | %-21s" % ("shape (M x K x N)", "", "fp32 (torch CPU)", "bf16 plain",
                                                             "split 3 terms", "split 6 terms", "split 9 terms"))
for name, M, K, N in [("layer3 conv2 3x3", 512, 2304, 256), ("layer4 conv2 3x3", 512, 4608, 512),
                      ("cls.0 3x3 4096->512", 256, 36864, 512)]:
    a = torch.randn(M, K).relu_() * 0.7 + 0.01 * torch.randn(M, K)      # post-ReLU activations
    b = torch.randn(K, N) * (2.0 / K) ** 0.5                             # kaiming-scaled weights
    ref = a.double() @ b.double()
    rows = [rel(a @ b, ref), rel((a.to(torch.bfloat16).float() @ b.to(torch.bfloat16).float()), ref),
            rel(gemm_split(a, b, THREE), ref), rel(gemm_split(a, b, SIX), ref), rel(gemm_split(a, b, NINE), ref)]
    print("%-28s %10s | " % (name, "max / rms") + " | ".join("%.2e / %.2e" % r for r in rows))
print("MFMA passes relative to one fp32 pass (bf16 MFMA = 16x the fp32-input rate on gfx950): 3 terms 0.19, 6 terms "
      "0.375, 9 terms 0.56 — plus the split itself (3 bf16 stores per operand element, once per tensor).")
