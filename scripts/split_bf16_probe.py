"""EXPERIMENT: the split-bf16 row GEMM (csrc/gemm_bf16split.hip) on PSPNet's cls.0 — time and error
next to the fp32 matrix-core kernel it would replace.  Never the reported configuration; prints a report, writes
gpurun_out/split_bf16_probe.json.

  A  the two cls.0 GEMMs of a batch-16 473x473 step (forward K 4096 -> 512, data gradient K 512 -> 4096, 16 Winograd
     positions x 14400 tiles), standalone: us per launch, TFLOP/s (fp32-equivalent), rms / max error against fp64
  B  PSPNet-101 473x473 eval logits against the CPU oracle with SEMSEG_SPLIT_BF16 = 0 / 3 / 6 (cls.0 only)
  C  in-situ backward check of cls.0 (PSPNet-101 473x473 batch 2): data-gradient error / CPU-fp32 error
  D  ms per train step (PSPNet-101 473x473 batch 16) with the flag 0 / 3 / 6 on cls.0, and on every eligible conv:
     3 = forward + data-gradient GEMMs of the Winograd convs; 6 = those plus forward / data gradient of every 1x1 conv
     (SP instances of conv_igemm_kernel); weight gradients always stay fp32 MFMA
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = torch.device("cuda:0")
OUT = {}


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def part_a():
    from semseg_amd import ops
    T, batch = 14400, 16
    res = {}
    for tag, K, Nout in (("cls.0 forward", 4096, 512), ("cls.0 data gradient", 512, 4096), ("layer4 conv2", 512, 512)):
        g = torch.Generator(device=DEV).manual_seed(K + Nout)
        A = torch.randn(batch, T, K, device=DEV, generator=g)
        Bt = torch.randn(batch, Nout, K, device=DEV, generator=g) / K ** 0.5
        C = torch.empty(batch, T, Nout, device=DEV)
        sl = slice(T - 512, T)                                   # includes the ragged last row tile
        ref = torch.stack([A[b, sl].double() @ Bt[b].double().t() for b in (0, batch - 1)])
        rms_ref = float(ref.pow(2).mean().sqrt())
        flops = 2.0 * batch * T * K * Nout
        rows = {}
        variants = [("fp32 mfma", None), ("bf16 x3 bk16", (2, 16)), ("bf16 x3 bk32", (2, 32)), ("bf16 x6 bk16", (3, 16)),
                    ("bf16 x6 igemm SP", "igemm")]
        for name, cfg in variants:
            if cfg == "igemm":
                def fn():
                    with ops.conv_split(True):
                        ops.gemm_rows_batched(A, K, T * K, Bt, Nout * K, C, Nout, T * Nout, T, K, Nout, batch)
            elif cfg is None:
                fn = lambda: ops.gemm_rows_batched(A, K, T * K, Bt, Nout * K, C, Nout, T * Nout, T, K, Nout, batch)
            else:
                fn = lambda cfg=cfg: ops.gemm_rows_batched_bf16split(A, K, T * K, Bt, Nout * K, C, Nout, T * Nout, T, K,
                                                                     Nout, batch, nsplit=cfg[0], bk=cfg[1])
            C.fill_(float("nan"))
            us = timed(fn)
            got = torch.stack([C[0, sl], C[batch - 1, sl]]).double()
            d = got - ref
            full_ok = bool(torch.isfinite(C).all())
            rows[name] = dict(us=us, tflops=flops / us * 1e-6, rms=float(d.pow(2).mean().sqrt()) / rms_ref,
                              max=float(d.abs().max()) / float(ref.abs().max()), finite=full_ok)
            print("A  %-20s %-16s %9.1f us  %7.1f TFLOP/s  rms %.2e  max %.2e  finite %s"
                  % (tag, name, us, rows[name]["tflops"], rows[name]["rms"], rows[name]["max"], full_ok), flush=True)
        # what torch's own fp32 matmul (hipBLASLt) makes of the same slice, for scale
        t32 = torch.stack([A[b, sl] @ Bt[b].t() for b in (0, batch - 1)]).double() - ref
        rows["torch fp32 matmul"] = dict(rms=float(t32.pow(2).mean().sqrt()) / rms_ref)
        print("A  %-20s torch fp32 matmul rms %.2e" % (tag, rows["torch fp32 matmul"]["rms"]), flush=True)
        res[tag] = rows
        del A, Bt, C
    OUT["A_gemm"] = res


def part_b():
    from semseg_amd import engine as E
    from oracle import segnet
    from test_model_gpu import build, inputs, rel
    m, sd = build("psp", 101, 150)
    x, _ = inputs(1, 473, 150)
    with torch.no_grad():
        ref = segnet.forward({k: v.clone() for k, v in sd.items()}, x, 101, "psp", training=False)
    res = {}
    for split in (0, 3, 6):
        E.SPLIT_BF16 = split
        mm, _ = build("psp", 101, 150)
        mm = mm.cuda().eval()
        out = mm(x.cuda())
        res[str(split)] = rel(out, ref)
        print("B  eval logits vs CPU oracle, SPLIT_BF16=%d: %.2e" % (split, res[str(split)]), flush=True)
        del mm
    E.SPLIT_BF16 = 0
    OUT["B_logits_rel_err"] = res


def part_c():
    from semseg_amd import engine as E
    import insitu
    from test_model_gpu import build, inputs

    class OnlyCls0(insitu.InsituChecker):
        def __call__(self, op):
            if op.kind == "conv" and self.names.get(op.ctx["m"]) == "cls.0":
                torch.cuda.synchronize()
                self._chk_conv(op)
            else:
                op.fn()

    res = {}
    x, y = inputs(2, 473, 150)
    for split in (0, 3, 6):
        E.SPLIT_BF16 = split
        m, _ = build("psp", 101, 150)
        m = m.cuda().train()
        eng = E.Engine(m, 2, 473, 473, True, "psp")
        chk = OnlyCls0(eng)
        eng.forward_train(x.cuda(), y.cuda(), 255)
        eng.tape_hook = chk
        eng.backward(torch.ones(1, device=DEV), torch.full((1,), 0.4, device=DEV))
        torch.cuda.synchronize()
        eng.tape_hook = None
        for kind, name, qty, mh, mc, rh, rc in chk.rows:
            res["%d %s" % (split, qty)] = dict(rms_hip=rh, rms_cpu32=rc, ratio=rh / rc, max_ratio=mh / mc)
            print("C  SPLIT_BF16=%d cls.0 %-16s rms hip %.2e cpu-fp32 %.2e ratio %.2f  max-abs ratio %.2f"
                  % (split, qty, rh, rc, rh / rc, mh / mc), flush=True)
        del eng, m, chk
    E.SPLIT_BF16 = 0
    OUT["C_insitu_cls0"] = res


def part_d():
    from semseg_amd import engine as E
    from semseg_amd.trainer import Trainer
    from model.pspnet import PSPNet
    res = {}
    g = torch.Generator().manual_seed(1000)
    x = torch.randn(16, 3, 473, 473, generator=g).to(DEV)
    y = torch.randint(0, 150, (16, 473, 473), generator=g).to(DEV)
    for split, layers, wk, wg in ((0, "cls.0", "standalone", False), (3, "cls.0", "standalone", False),
                                  (6, "cls.0", "standalone", False), (3, "all", "standalone", False),
                                  (6, "all", "standalone", False), (6, "all", "igemm", False),
                                  (6, "all", "standalone", True), (0, "cls.0", "standalone", False)):
        E.SPLIT_BF16 = split
        E.SPLIT_LAYERS = [layers]
        E.SPLIT_WINO_KERNEL = wk
        E.SPLIT_WGRAD = wg
        torch.manual_seed(0)
        model = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False).to(DEV).train()
        tr = Trainer(model, base_lr=0.01, momentum=0.9, weight_decay=1e-4, aux_weight=0.4, sync_bn=True)
        losses = []
        for _ in range(3):
            losses.append(tr.step(x, y, 0.01)[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            losses.append(tr.step(x, y, 0.01)[1])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        lv = [float(l) for l in losses]
        nsp = sum(1 for e in tr.engines.values() for c in e.convs.values() if c is not None and c.split)
        nw = sum(1 for e in tr.engines.values() for c in e.convs.values() if c is not None and c.split_w)
        res.setdefault("%d %s %s wgrad=%d" % (split, layers, wk, wg), []).append(
            dict(ms=ms, loss_after_11_steps=lv[-1], convs_split=nsp, convs_split_wgrad=nw))
        print("D  SPLIT_BF16=%d layers=%s wino-kernel=%s (%d convs fwd/dgrad, %d wgrad)  %.1f ms/step  loss after 11 steps %.5f"
              % (split, layers, wk, nsp, nw, ms, lv[-1]), flush=True)
        del tr, model
        torch.cuda.empty_cache()
    E.SPLIT_BF16, E.SPLIT_LAYERS, E.SPLIT_WINO_KERNEL, E.SPLIT_WGRAD = 0, ["cls.0"], "standalone", True
    OUT["D_step_ms"] = res


if __name__ == "__main__":
    parts = sys.argv[1:] or ["a", "b", "c", "d"]
    for p in parts:
        try:
            {"a": part_a, "b": part_b, "c": part_c, "d": part_d}[p]()
        except Exception as e:                                  # keep the other parts' numbers
            import traceback
            traceback.print_exc()
            OUT["error_" + p] = repr(e)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "split_bf16_probe.json"), "w") as f:
        json.dump(OUT, f, indent=1)
