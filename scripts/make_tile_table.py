"""Regenerate semseg_amd/tile_table.json on an MI355X: for every forward / data-gradient shape of the BASELINE.json
configurations (PSPNet-101 473^2 at per-GPU batch 16 / 8 / 4 / 2, PSANet-101 465^2 at 16 / 2, PSPNet-101 713^2 at 2)
time the four tile shapes (128 x 128, 128 x 64, 64 x 128, 64 x 64; with --split also code 2128 = the 256 x 128 bf16x3 GEMM
kernel for the forward of 1x1 stride-1 convs) on real operands (semseg_amd.ops._tuned_tile, 3
interleaved rounds of 3 launches, device idle) and keep 128 x 128 unless another shape wins by >= 3 %.  The table is committed; nothing times tiles at run time.

    SEMSEG_TILE_TUNE=1 python scripts/make_tile_table.py [out.json]        (GPU box; copy the result into semseg_amd/)
    ... make_tile_table.py --split [out.json]   the same for the SEMSEG_ARITH_BF16X3 kernel instances (keys "...|sp",
                                                semseg_amd/tile_table_sp.json; every configuration above) — the engine default
    (without --split the engines are built with exact fp32 arithmetic: the table of the SEMSEG_ARITH=f32 path)
"""
import json
import os
import sys

os.environ["SEMSEG_TILE_TUNE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from semseg_amd import ops  # noqa: E402
from semseg_amd.trainer import Trainer  # noqa: E402

CONFIGS = [("psp", 101, 473, 150, b) for b in (16, 8, 4, 2)] + [("psa", 101, 465, 150, b) for b in (16, 2)] + \
          [("psp", 101, 713, 19, 2)]

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--split"]
    SPLIT = "--split" in sys.argv[1:]
    out = args[0] if args else os.path.join(ROOT, "gpurun_out", "tile_table_sp.json" if SPLIT else "tile_table.json")
    from semseg_amd import engine as E
    E.set_arith("bf16x3" if SPLIT else "f32")
    for k in [k for k in ops.TILE_CHOICE if k.endswith("|sp") == SPLIT]:      # measure everything of this table afresh
        del ops.TILE_CHOICE[k]
    for arch, layers, size, classes, bs in CONFIGS:
        torch.manual_seed(0)
        if arch == "psp":
            from model.pspnet import PSPNet
            m = PSPNet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
        else:
            from model.psanet import PSANet
            m = PSANet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
        m = m.cuda().train()
        tr = Trainer(m, sync_bn=False)
        x = torch.randn(bs, 3, size, size, device="cuda")
        y = torch.randint(0, classes, (bs, size, size), device="cuda")
        n0 = len(ops.TILE_CHOICE)
        tr.step(x, y)
        torch.cuda.synchronize()
        print("%s%d %d^2 bs %d: %d new shapes" % (arch, layers, size, bs, len(ops.TILE_CHOICE) - n0), flush=True)
        del tr, m, x, y
        torch.cuda.empty_cache()
    tiles = dict(sorted((k, v) for k, v in ops.TILE_CHOICE.items() if k.endswith("|sp") == SPLIT))
    doc = {"generated_by": "scripts/make_tile_table.py (3 x 3 launches per tile shape; 128 x 128 unless another wins by >= 3 %, codes 2128 / 3128 by >= 2 %)",
           "tile_codes": "128 = 128x128, 64 = 128x64, 1128 = 64x128, 1064 = 64x64 (rows x columns); 2128 = forward of a 1x1 "
                         "stride-1 conv (or its data gradient) on the 256x128 bf16x3 GEMM kernel (gemm_bf16split.hip), 3128 = the same kernel with 128x128 tiles",
           "configs": ["%s%d_%d_c%d_bs%d" % c for c in CONFIGS],
           "tiles": tiles,
           "ms_per_tile_code": {k: {str(c): t for c, t in v.items()} for k, v in sorted(ops.TILE_TIMES.items())}}
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=False)
    print("wrote %s: %d shapes, by tile code %s" % (out, len(tiles), {c: sum(1 for v in tiles.values() if v == c) for c in ops.TILE_CODES + ops.SPLIT_GEMM_CODES}))
