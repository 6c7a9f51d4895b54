#!/usr/bin/env python
"""bench.py — images/sec of the PSPNet-101 train step (473x473, global batch 16, 150 classes, fp32)
on N MI355X GPUs of one node, the metric BASELINE.json names.

Arithmetic: fp32 tensors everywhere; the conv GEMMs form their fp32 products per launch either from three-way split
bf16 pieces (SEMSEG_ARITH_BF16X3, the default: `value`, `dtype` says so) or on the fp32 matrix-core instruction
(SEMSEG_ARITH_F32: the `exact_fp32` object of the same JSON line, same protocol, same step count).

One "step" = the loop body of the reference's tool/train.py:269-276 on a synthetic batch that is
already resident in HBM: forward (SyncBN), loss = main + 0.4*aux, backward, gradient all-reduce
(N>1), SGD(momentum 0.9, wd 1e-4, two lr groups).  Launch for N>1:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
ISO_STEPS = 2
# per-kernel HBM bytes / MFMA-busy from the rocprofv3 --pmc passes of this round, one file per profiled configuration: each file
# names the workload it was collected on (`workload` = config.workload of the bench line of the same passes) and _attach_pmc only
# uses a file whose workload IS the running one — bytes of a batch-16 launch say nothing about a batch-2 launch
PMC_FILES = ("r06_pmc_per_kernel.json", "r06_bs2_pmc_per_kernel.json")


def shape_name(size, classes):
    """Which BASELINE.json dataset shape (size, classes) is, for config.workload."""
    if classes == 150 and size in (473, 465):
        return "ADE20K-shape"
    if classes == 19 and size in (713, 705):
        return "Cityscapes-shape"
    return "custom-shape"


def conv_flops_per_image(model, size):
    """Algorithmic conv (+ PSA bmm) FLOPs (2*MAC) of one forward, per image (477.4 GF for
    PSPNet101 @473, 487.2 + 1.66 GF for PSANet101 @465)."""
    from semseg_amd.ops import conv_out
    import torch.nn as nn
    # spatial size seen by each conv: replay the stride structure of the trunk
    total = 0
    s = size
    hw = {}
    l0 = model.layer0
    s0 = conv_out(s, 3, 2, 1, 1)
    hw[l0[0]] = (s, s0)
    hw[l0[3]] = (s0, s0)
    hw[l0[6]] = (s0, s0)
    sp = conv_out(s0, 3, 2, 1, 1)
    cur = sp
    for layer in (model.layer1, model.layer2, model.layer3, model.layer4):
        for blk in layer:
            st = blk.conv2.stride[0]
            out = conv_out(cur, 3, st, blk.conv2.padding[0], blk.conv2.dilation[0])
            hw[blk.conv1] = (cur, cur)
            hw[blk.conv2] = (cur, out)
            hw[blk.conv3] = (out, out)
            if blk.downsample is not None:
                hw[blk.downsample[0]] = (cur, out)
            cur = out
    feat = cur
    bmm = 0.0
    if hasattr(model, "psa") and getattr(model, "use_psa", True):
        # PSA head (model/psanet.py:53-98): reduce convs on the trunk map, attention convs and proj on the shrunk map,
        # plus the point-affinity contraction torch.bmm(x [C, hw], y [hw, hw]) per branch (psanet.py:90-91)
        psa = model.psa
        sf = psa.shrink_factor
        small = (feat - 1) // sf + 1 if sf != 1 else feat
        for name, m in psa.named_modules():
            if isinstance(m, nn.Conv2d):
                hw[m] = (feat, feat) if name.startswith("reduce") else (small, small)
        nb = 2 if psa.psa_type == 2 else 1
        bmm = nb * 2.0 * (small * small) ** 2 * psa.reduce[0].weight.shape[0]
    for m in model.modules():
        if isinstance(m, nn.Conv2d) and m not in hw:
            hw[m] = (feat, feat)
    if hasattr(model, "ppm") and getattr(model, "use_ppm", True):
        for f in model.ppm.features:
            b = f[0].output_size
            b = b if isinstance(b, int) else b[0]
            hw[f[1]] = (b, b)
    for m, (_, o) in hw.items():
        co, ci, r, s_ = m.weight.shape
        total += 2.0 * o * o * co * ci * r * s_
    total += bmm
    first = 2.0 * s0 * s0 * 64 * 3 * 9
    return total, first


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(layers, classes, size, iters=2, arch="psp"):
    """The reference's arithmetic on the host cores: oracle/segnet.py (bit-identical to the imported
    reference, see tests/golden/make_golden.py) forward + backward + SGD, batch 2."""
    from oracle import segnet
    from model.pspnet import PSPNet
    torch.manual_seed(0)
    cores = usable_cores()
    torch.set_num_threads(cores)
    psa_cfg = None
    if arch == "psp":
        m = PSPNet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
    else:
        from model.psanet import PSANet
        m = PSANet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
        psa_cfg = dict(psa_type=m.psa.psa_type, compact=m.psa.compact, shrink_factor=m.psa.shrink_factor,
                       mask_h=m.psa.mask_h, mask_w=m.psa.mask_w, normalization_factor=m.psa.normalization_factor,
                       psa_softmax=m.psa.psa_softmax)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    opt = torch.optim.SGD(list(params.values()), lr=0.01, momentum=0.9, weight_decay=1e-4)
    B = 2
    x = torch.randn(B, 3, size, size)
    y = torch.randint(0, classes, (B, size, size))
    times = []
    for it in range(iters + 1):
        t0 = time.time()
        _, ml, al = segnet.forward(sd, x, layers, arch, training=True, y=y, psa_cfg=psa_cfg)
        loss = ml + 0.4 * al
        opt.zero_grad()
        loss.backward()
        opt.step()
        times.append(time.time() - t0)
        if it >= 1 and sum(times) > 45.0:  # keep the default run within minutes on slow hosts
            break
    iters = len(times) - 1
    t = sum(times[1:]) / iters
    return {"value": round(B / t, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "PS%sNet%d %dx%d train step (fwd+bwd+SGD) batch %d, %d timed iterations after 1 warm-up, "
                      "torch CPU fp32 via oracle/segnet.py" % ("P" if arch == "psp" else "A", layers, size, size, B,
                                                               iters)}


def _tile_summary():
    """The committed per-shape tile-width table every process runs with (semseg_amd/tile_table.json)."""
    import hashlib
    from semseg_amd import ops
    if not os.path.exists(ops.TILE_TABLE_PATH):
        return "none (static default: 128-wide tiles)"
    return "semseg_amd/tile_table.json: %d shapes, sha256 %s" % (
        len(ops.TILE_CHOICE), hashlib.sha256(open(ops.TILE_TABLE_PATH, "rb").read()).hexdigest()[:12])


def module_path_step_time(args, dev, world, rank, B, steps, warmup):
    """The reference's own loop body (tool/train.py:269-276, 299-304) on the drop-in nn.Module API: torch.optim.SGD
    over the eight parameter groups of train.py:125-140, nn.SyncBatchNorm conversion + DistributedDataParallel under
    torch.distributed (train.py:142,157), autograd driving the HIP engine through semseg_amd.module_base._NetFunction.
    Returns seconds per step (max over ranks)."""
    import torch.nn as nn
    from semseg_amd.trainer import poly_learning_rate
    torch.manual_seed(0)
    if args.arch == "psp":
        from model.pspnet import PSPNet
        model = PSPNet(layers=args.layers, classes=args.classes, zoom_factor=8, pretrained=False)
        modules_new = [model.ppm, model.cls, model.aux]
    else:
        from model.psanet import PSANet
        model = PSANet(layers=args.layers, classes=args.classes, zoom_factor=8, pretrained=False)
        modules_new = [model.psa, model.cls, model.aux]
    modules_ori = [model.layer0, model.layer1, model.layer2, model.layer3, model.layer4]
    base_lr, index_split = 0.01, 5
    params_list = [dict(params=m.parameters(), lr=base_lr) for m in modules_ori]
    params_list += [dict(params=m.parameters(), lr=base_lr * 10) for m in modules_new]
    optimizer = torch.optim.SGD(params_list, lr=base_lr, momentum=0.9, weight_decay=1e-4)
    distributed = dist.is_initialized()
    if distributed:
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = torch.nn.parallel.DistributedDataParallel(model.to(dev), device_ids=[dev.index])
    else:
        model = model.to(dev)
    model.train()
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(B, 3, args.size, args.size, generator=g).to(dev)
    y = torch.randint(0, args.classes, (B, args.size, args.size), generator=g).to(dev)
    max_iter = steps + warmup + 1

    def one(it):
        output, main_loss, aux_loss = model(x, y)
        loss = main_loss + 0.4 * aux_loss
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lr = poly_learning_rate(base_lr, it, max_iter)
        for index in range(0, index_split):
            optimizer.param_groups[index]['lr'] = lr
        for index in range(index_split, len(optimizer.param_groups)):
            optimizer.param_groups[index]['lr'] = lr * 10
        return main_loss

    for it in range(warmup):
        one(it)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for it in range(warmup, warmup + steps):
        ml = one(it)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(ml.item())
    del model, optimizer
    torch.cuda.empty_cache()
    return dt / steps, loss_val


PEAK_BF16_MFMA_TFLOPS = 2500.0   # same guide: v_mfma_f32_32x32x16_bf16, dense
# SEMSEG_ARITH_BF16X3 spends six bf16 matrix-core products per fp32 product: the peak of a bf16x3 kernel in fp32-equivalent
# FLOPs is the dense bf16 peak / 6 ("no quoting 197 TF against 157.3")
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def family_peak(family):
    """Matrix-core peak (fp32-equivalent TFLOP/s) of a kernel family by the arithmetic its label names."""
    return PEAK_BF16X3_TFLOPS if ("SP3" in family or "bf16x3" in family) else PEAK_F32_MFMA_TFLOPS


def build_model(args):
    torch.manual_seed(0)  # identical initial weights on every rank (what DDP's broadcast achieves)
    if args.arch == "psp":
        from model.pspnet import PSPNet
        return PSPNet(layers=args.layers, classes=args.classes, zoom_factor=8, pretrained=False)
    from model.psanet import PSANet
    return PSANet(layers=args.layers, classes=args.classes, zoom_factor=8, pretrained=False)


def trainer_leg(args, dev, world, rank, B, steps, warmup, arith, kernel_timing, dist_on):
    """`steps` timed Trainer steps (barrier + synchronize on both sides, max over ranks) with the conv GEMMs in `arith`
    ("bf16x3" | "f32"), then — with kernel_timing — ISO_STEPS more steps with every kernel on ONE stream for the per-family
    HIP-event durations the roofline uses.  Returns a dict; the Trainer and its engine are released before returning."""
    from semseg_amd.trainer import Trainer, poly_learning_rate
    from semseg_amd import engine as E
    old = E.set_arith(arith)
    try:
        model = build_model(args).to(dev).train()
        tr = Trainer(model, base_lr=0.01, momentum=0.9, weight_decay=1e-4, aux_weight=0.4, sync_bn=True)
        g = torch.Generator().manual_seed(1000 + rank)
        x = torch.randn(B, 3, args.size, args.size, generator=g).to(dev)
        y = torch.randint(0, args.classes, (B, args.size, args.size), generator=g).to(dev)
        max_iter = steps + warmup + 1 + 2 * ISO_STEPS
        it = 0
        for _ in range(warmup):
            tr.step(x, y, poly_learning_rate(0.01, it, max_iter))
            it += 1
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            _, main_loss, aux_loss = tr.step(x, y, poly_learning_rate(0.01, it, max_iter))
            it += 1
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.time() - t0
        if dist_on:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        out = {"sec": dt / steps, "loss": float(main_loss.item()), "arith": arith}
        # who sequenced the launches of the timed steps: the C-side replay of a recorded step (csrc/plan.hip) or Python
        replayed = sum(getattr(e, "_plan_replays", 0) for e in tr.engines.values())
        if dist_on:      # which SyncBN exchange the job took, and why (peer-memory kernel after its start-up self-test, or RCCL)
            from semseg_amd import syncbn_xchg
            out["syncbn_exchange"] = syncbn_xchg.DECISION.get(dev.index, (None, "no SyncBN exchange was issued"))[1]
        out["step_driver"] = {"timed_steps_replayed_from_C": min(replayed, steps),
                              "log": tr.plan_log[-1:] if tr.use_plan else ["SEMSEG_STEP_PLAN=0: launch by launch from Python"]}
        for e in tr.engines.values():
            e.check_labels()
            out["n_sync"] = getattr(e, "syncbn_collectives_per_step", None)
            out["two_stream_backward"] = bool(e.side_wgrad)
            out["convs_bf16x3"] = sum(1 for c in e.convs.values() if c is not None and c.arith == 3)
        # The timed region above carries NO instrumentation (rounds 1-3 recorded two HIP events around every matrix-core
        # launch inside it: ~1 ms per step at batch 16, ~3 ms at per-GPU batch 2).  Kernel timing runs on extra steps:
        # (1) ISO_STEPS steps in the configuration of the timed region (two streams) -> kernel_families_in_step;
        # (2) the serialized leg (every rank runs both: the steps contain the SyncBN / gradient collectives): ISO_STEPS more
        # steps with all kernels on one stream, HIP-event timed on rank 0 -> per-kernel rates that are not inflated by
        # the side-stream concurrency.
        if kernel_timing:
            kt = E.KernelTimer() if rank == 0 else None
            for e in tr.engines.values():
                e.ktimer = kt
            for _ in range(ISO_STEPS):
                tr.step(x, y, poly_learning_rate(0.01, it, max_iter))
                it += 1
            torch.cuda.synchronize()
            kt_iso = E.KernelTimer() if rank == 0 else None
            saved = [(e, e.side_wgrad, e.hipri_main) for e in tr.engines.values()]
            for e, _, _ in saved:
                e.side_wgrad, e.hipri_main, e.ktimer = False, False, kt_iso
            for _ in range(ISO_STEPS):
                tr.step(x, y, poly_learning_rate(0.01, it, max_iter))
            torch.cuda.synchronize()
            for e, sw, hp in saved:
                e.side_wgrad, e.hipri_main, e.ktimer = sw, hp, None
            if rank == 0:
                out["exec_flops"] = kt_iso.mfma_flops() / ISO_STEPS
                out["kernel_families"] = kt_iso.summary()
                out["kernel_families_in_step"] = kt.summary()
                fam = kt_iso.dominant_family()
                roof = kt_iso.roofline(family_peak(fam), family=fam)
                ins = kt.roofline(family_peak(fam), family=fam)
                roof["in_step"] = {k: ins[k] for k in ("achieved", "frac", "avg_launch_us", "launches")}
                out["roofline"] = roof
        del tr, model
        torch.cuda.empty_cache()
    finally:
        E.set_arith(old)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=5)      # >= 4: two launch-by-launch steps + the two recorded steps of the step plan
    ap.add_argument("--layers", type=int, default=101)
    ap.add_argument("--size", type=int, default=473)
    ap.add_argument("--classes", type=int, default=150)
    ap.add_argument("--global-batch", type=int, default=16)
    ap.add_argument("--arch", default="psp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 leg reported beside the value")
    ap.add_argument("--no-experiments", dest="no_exact", action="store_true", help=argparse.SUPPRESS)   # round-3 spelling
    ap.add_argument("--arith", default=None, choices=["bf16x3", "f32"],
                    help="arithmetic of the headline leg (default: the engine default, bf16x3 unless SEMSEG_ARITH says f32)")
    ap.add_argument("--path", default="trainer", choices=["trainer", "module"],
                    help="trainer: the fused loop body (semseg_amd.Trainer) is the timed value and the drop-in nn.Module "
                         "+ torch.optim.SGD loop is timed beside it (module_path); module: the other way round")
    ap.add_argument("--module-steps", type=int, default=6, help="timed steps of the path reported beside the value")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("SEMSEG_FORCE_DIST") == "1" and "RANK" in os.environ
    dist_on = world > 1 or force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    assert args.global_batch % world == 0
    B = args.global_batch // world

    from semseg_amd import engine as E
    if args.arith is not None:
        E.set_arith(args.arith)
    arith = E.arith_name()
    fwd_flops, first_flops = conv_flops_per_image(build_model(args), args.size)

    primary_trainer = args.path == "trainer"
    steps_t = args.steps if primary_trainer else args.module_steps
    warm_t = args.warmup if primary_trainer else 2
    leg = trainer_leg(args, dev, world, rank, B, steps_t, warm_t, arith, not args.no_kernel_timing, dist_on)
    sec_trainer, loss_trainer = leg["sec"], leg["loss"]

    # The other path: the reference's unchanged loop body on the nn.Module API (torch.optim.SGD, autograd, DDP).
    # The second leg is a single-GPU comparison; in a multi-rank launch only the path that was asked for runs (a failure
    # in an extra leg on one rank would hang the others in a collective).
    sec_module = loss_module = None
    if (args.module_steps > 0 and world == 1) or not primary_trainer:
        sec_module, loss_module = module_path_step_time(args, dev, world, rank, B,
                                                        args.module_steps if primary_trainer else args.steps,
                                                        2 if primary_trainer else args.warmup)
    if primary_trainer:
        dt, loss_val = sec_trainer * args.steps, loss_trainer
    else:
        dt, loss_val = sec_module * args.steps, loss_module

    # The exact-fp32 configuration (every product on v_mfma_f32_32x32x2_f32: the arithmetic of rounds 1-3), timed by the
    # same protocol with the same number of steps in the same process, so that both figures are driver-timed.
    exact = None
    if world == 1 and not args.no_exact and arith != "f32":
        try:
            ex = trainer_leg(args, dev, world, rank, B, args.steps, args.warmup, "f32", not args.no_kernel_timing, dist_on)
            exact = {"what": "the same Trainer step with SEMSEG_ARITH_F32 on every conv GEMM (SEMSEG_ARITH=f32): exact fp32 "
                             "products on the fp32-input matrix-core instruction, the headline configuration of rounds 1-3",
                     "dtype": "f32", "ms_per_step": round(ex["sec"] * 1e3, 3),
                     "images_per_sec": round(args.global_batch / ex["sec"], 3), "steps": args.steps, "warmup": args.warmup,
                     "final_main_loss": round(ex["loss"], 5), "roofline": ex.get("roofline")}
            if "exec_flops" in ex:
                exact["executed_mfma_tflop_per_step"] = round(ex["exec_flops"] / 1e12, 3)
                exact["whole_step_frac_of_f32_mfma_peak"] = round(ex["exec_flops"] / ex["sec"] / 1e12 /
                                                                  PEAK_F32_MFMA_TFLOPS, 4)
                exact["kernel_families"] = ex["kernel_families"]
        except Exception as e:                       # the comparison leg never takes the bench line down
            exact = {"error": repr(e)}

    if rank == 0:
        ips = args.global_batch * args.steps / dt
        step_flops = (3.0 * fwd_flops - first_flops) * args.global_batch
        dtype = ("f32 (3xbf16-split products, fp32 accumulate)" if arith == "bf16x3" else "f32")
        out = {
            "metric": "images/sec (train step) %s-%d %dx%d bs=%d" % ("PSPNet" if args.arch == "psp" else "PSANet",
                                                                 args.layers, args.size, args.size,
                                                                 args.global_batch),
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload_name(args, B),
                       "parallelism": "dp%d" % world,
                       "path": ("semseg_amd.Trainer (fused loop body of tool/train.py:269-276)" if primary_trainer else
                                "nn.Module drop-in + torch.optim.SGD (tool/train.py:269-276 unchanged)"),
                       "arith": ("SEMSEG_ARITH_BF16X3 per launch on %d of the network's conv layers (forward, data gradient, "
                                 "weight gradient; include/semseg_hip.h): fp32 operands in HBM, each cut in flight into "
                                 "three bf16 pieces carrying all 24 mantissa bits, six cross products on "
                                 "v_mfma_f32_32x32x16_bf16, fp32 accumulation; everything that is not a conv GEMM is "
                                 "fp32" % leg.get("convs_bf16x3", 0)) if arith == "bf16x3" else
                                "SEMSEG_ARITH_F32: exact fp32 products on v_mfma_f32_32x32x2_f32",
                       "two_stream_backward": leg.get("two_stream_backward"),
                       "step_driver": leg.get("step_driver"),
                       "syncbn_exchange": leg.get("syncbn_exchange", "single process: no exchange"),
                       "tile_table": _tile_summary()},
            "final_main_loss": round(loss_val, 5),
            # direct-convolution FLOPs of the step (SURVEY.md section 8d / BASELINE.md section 4: 3 x forward - first conv's
            # data gradient) — the work the metric is defined on; the stride-1 3x3 convs with >= 128 channels EXECUTE
            # 1 / 2.25 of their share (Winograd F(2x2,3x3)), so this rate can exceed what the matrix cores ran
            "algorithmic_tflop_per_step": round(step_flops / 1e12, 3),
            "algorithmic_tflops_over_f32_mfma_peak": round(step_flops / (dt / args.steps) / 1e12 / world /
                                                           PEAK_F32_MFMA_TFLOPS, 4),
        }
        if "exec_flops" in leg and primary_trainer:
            # executed matrix-core FLOPs (fp32-equivalent: one multiply-add = 2, however many bf16 products form it) are
            # measured on the Trainer leg, so they are only reported against the Trainer leg's step time
            rate = leg["exec_flops"] / (dt / args.steps) / 1e12
            out["executed_mfma_tflop_per_step"] = round(leg["exec_flops"] * world / 1e12, 3)
            out["whole_step_executed_tflops"] = round(rate, 2)
            out["whole_step_frac_of_f32_mfma_peak"] = round(rate / PEAK_F32_MFMA_TFLOPS, 4)
            if arith == "bf16x3":
                out["whole_step_frac_of_bf16x3_peak"] = round(rate / PEAK_BF16X3_TFLOPS, 4)
        other = sec_module if primary_trainer else sec_trainer
        if other is not None:
            out["module_path" if primary_trainer else "trainer_path"] = {
                "what": ("tool/train.py:269-276 loop body on the nn.Module API: torch.optim.SGD (8 param groups), "
                         "autograd, SyncBN convert + DDP under torch.distributed" if primary_trainer
                         else "semseg_amd.Trainer fused step"),
                "ms_per_step": round(other * 1e3, 3), "images_per_sec": round(args.global_batch / other, 3),
                "steps": args.module_steps, "final_main_loss": round(loss_module if primary_trainer else loss_trainer, 5)}
        if exact is not None:
            out["exact_fp32"] = exact
        if leg.get("n_sync") is not None:
            out["syncbn_collectives_per_step"] = leg["n_sync"]
        roof = leg.get("roofline")
        if roof is not None:
            # roofline of the dominant kernel family = the serialized leg (the kernel has the chip to itself);
            # what the same family shows inside the timed region, where it may share the chip, is reported next to it
            roof["peak_note"] = ("bf16x3 kernels: dense bf16 matrix-core peak %.0f TFLOP/s / 6 products = %.1f TFLOP/s "
                                 "fp32-equivalent; fp32-instruction kernels: %.1f" % (PEAK_BF16_MFMA_TFLOPS,
                                                                                     PEAK_BF16X3_TFLOPS,
                                                                                     PEAK_F32_MFMA_TFLOPS))
            roof["measured"] = ("HIP events on the launch stream over %d steps run right after the timed region with "
                                "every kernel on ONE stream (in_step: the same family over %d more steps in the two-stream "
                                "configuration of the timed region; the timed region itself carries no events)"
                                % (ISO_STEPS, ISO_STEPS))
            _attach_pmc(roof, args, B)
            out["roofline"] = roof
            out["kernel_families"] = leg["kernel_families"]
            out["kernel_families_in_step"] = leg["kernel_families_in_step"]
            if exact is not None and isinstance(exact.get("roofline"), dict):
                _attach_pmc(exact["roofline"], args, B)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.layers, args.classes, args.size, arch=args.arch)
        print(json.dumps(out))
        sys.stdout.flush()
    if dist_on:
        dist.destroy_process_group()


def workload_name(args, B):
    return ("PS%sNet%d %s %dx%d, %d classes, global batch %d (per-GPU %d), train step fwd+loss+bwd+SGD, SyncBN, random-init weights"
            % ("P" if args.arch == "psp" else "A", args.layers, shape_name(args.size, args.classes), args.size, args.size,
               args.classes, args.global_batch, B))


def _attach_pmc(roof, args, B):
    """HBM traffic / matrix-pipe busy fraction of that kernel from the PMC passes committed under profiles/ (rocprofv3 cannot
    run inside this process; the file records the exact commands and corrections).  Only from a file collected on THIS network,
    size, class count and PER-GPU batch on gfx950; otherwise `traffic` stays null."""
    import re
    want = re.sub(r"global batch \d+ ", "", workload_name(args, B))      # the per-GPU batch decides a launch's bytes
    for name in PMC_FILES:
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if re.sub(r"global batch \d+ ", "", pmc.get("workload", "")) != want or pmc.get("arch", "gfx950") != "gfx950":
            continue
        key = roof["kernel"].split("+")[0].split("(")[0].replace(" ", "")
        for k, v in pmc["kernels"].items():
            if k.replace(" ", "") == key:
                roof["traffic"] = v["hbm_bytes_per_launch"]
                roof["traffic_unit"] = "bytes/launch (PMC FETCH_SIZE*2 + WRITE_SIZE, profiles/%s: %s)" % (name, pmc["workload"])
                roof["mfma_busy_frac_pmc"] = v.get("mfma_busy_frac")
        return


if __name__ == "__main__":
    main()
