"""Static-shape executor of the PSPNet / PSANet forward + backward on the HIP kernel library.

The nn.Module tree (model/pspnet.py, model/psanet.py in this repo) only *holds parameters* under the
reference's state-dict names; this engine walks that tree once per (batch, H, W, mode), owns every
activation / gradient buffer (NHWC fp32, explicit channel stride), and issues the C-ABI kernels
(include/semseg_hip.h) on the current HIP stream.  Forward appends backward closures to a tape; the
backward pass replays it in reverse.  No torch compute op is on the path (torch = device memory,
streams, the seed of the Dropout2d mask generator, torch.distributed for the SyncBN / gradient all-reduce).

Reference call structure mirrored here: model/pspnet.py:80-105 (PSPNet.forward),
model/resnet.py:74-94 (Bottleneck.forward), model/pspnet.py:21-26 (PPM.forward),
model/psanet.py:53-98 (PSA.forward).
"""
import os

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from ._lib import debug

F32 = torch.float32
F64 = torch.float64


class KernelTimer:
    """HIP-event timing of the matrix-core kernels on the stream they are launched on (torch's current
    stream).  Families are named after the kernel template instantiation rocprofv3 reports."""

    def __init__(self):
        self.rec = []

    def span(self, family, flops):
        """flops > 0: matrix-core work (algorithmic FLOPs executed by the launch); flops < 0: an HBM-bound kernel,
        -flops = algorithmic bytes it moves."""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.rec.append((family, flops, s, e))
        return s, e

    def _collect(self):
        torch.cuda.synchronize()
        fam = {}
        for family, flops, s, e in self.rec:
            d = fam.setdefault(family, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += s.elapsed_time(e) * 1e-3
        return fam

    def summary(self):
        out = {}
        for k, (n, fl, t) in sorted(self._collect().items(), key=lambda kv: -kv[1][2]):
            out[k] = {"launches": n, "total_ms": round(t * 1e3, 3), "avg_us": round(t / n * 1e6, 2)}
            if fl >= 0:
                out[k]["tflops"] = round(fl / t / 1e12, 2) if t > 0 else None
            else:
                out[k]["hbm_tb_per_s_algorithmic"] = round(-fl / t / 1e12, 2) if t > 0 else None
        return out

    def mfma_flops(self):
        """FLOPs executed on the matrix cores over everything recorded (Winograd GEMMs count what they execute)."""
        return sum(fl for _, fl, _, _ in self.rec if fl > 0)

    def dominant_family(self):
        """The matrix-core family with the largest total time."""
        fam = {k: v for k, v in self._collect().items() if v[1] > 0}
        return max(fam.items(), key=lambda kv: kv[1][2])[0]

    def roofline(self, peak_tflops, family=None):
        fam = self._collect()
        fam = {k: v for k, v in fam.items() if v[1] > 0}      # matrix-core families only
        if family is not None:
            k, (n, fl, t) = family, fam[family]
        else:
            k, (n, fl, t) = max(fam.items(), key=lambda kv: kv[1][2])
        ach = fl / t / 1e12
        return {"kernel": k, "bound": "mfma", "achieved": round(ach, 2), "peak": peak_tflops,
                "unit": "TFLOP/s", "frac": round(ach / peak_tflops, 4), "traffic": None,
                "launches": n, "avg_launch_us": round(t / n * 1e6, 2),
                "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3)}


class TapeOp:
    """One backward step of the tape: `fn()` launches it; `kind` + `ctx` name the operands so a test hook
    (Engine.tape_hook) can snapshot them around the launch and re-derive the result independently."""
    __slots__ = ("kind", "fn", "ctx")

    def __init__(self, kind, fn, ctx):
        self.kind, self.fn, self.ctx = kind, fn, ctx


class Act:
    """NHWC activation view: `data` starts at the first valid channel; `ld` = channel stride."""
    __slots__ = ("data", "N", "H", "W", "C", "ld", "grad", "ginit", "name", "bnsrc", "fuse_ok", "pending", "bn_reduced",
                 "bits")

    def __init__(self, data, N, H, W, C, ld, name=""):
        self.data, self.N, self.H, self.W, self.C, self.ld = data, N, H, W, C, ld
        self.grad = None
        self.ginit = False
        self.name = name
        # fused BatchNorm-backward reduction (Engine._conv_bwd): which BatchNorm(s) produced this activation, whether
        # all of its consumers are convs / residual adds (set by Engine.bottleneck), how many gradient contributions
        # are still outstanding in backward, and whether the last one already did the reduction
        self.bnsrc = None
        self.fuse_ok = False
        self.bits = None       # ReLU mask of a BatchNorm+ReLU output as bits [M][C / 32] (bn_act, training)
        self.pending = 0
        self.bn_reduced = False

    @property
    def M(self):
        return self.N * self.H * self.W

    def slice(self, c0, C):
        a = Act(self.data[..., c0:], self.N, self.H, self.W, C, self.ld, self.name + "[%d:]" % c0)
        return a


WINO_HBM = "wino_input / wino_output / wino_dy_wgrad / wino_filter_grad kernels (Winograd transforms, HBM-bound)"
WINOGRAD = debug("winograd", "1") != "0"   # 0: every 3x3 conv on the direct implicit-GEMM kernels (A/B)
# Arithmetic of the matrix-core products of every conv GEMM of an engine (include/semseg_hip.h):
#   "bf16x3" (default)  SEMSEG_ARITH_BF16X3: each fp32 operand cut in flight into three bf16 pieces (all 24 mantissa bits),
#                       six cross products on the bf16 matrix-core instruction, fp32 accumulation — fp32-grade by every
#                       parity criterion of this repo (in situ: at or below the fp32 instruction's error)
#   "f32"               SEMSEG_ARITH_F32: exact fp32 products everywhere (the configuration of rounds 1-3)
# Read when an Engine is BUILT (SEMSEG_ARITH in the environment, or set_arith() before the first forward of a model /
# Trainer); it is handed to the C ABI per launch, there is no process-wide switch in the library.
_ARITH_NAMES = {"f32": ops.ARITH_F32, "bf16x3": ops.ARITH_BF16X3}
ARITH = _ARITH_NAMES[os.environ.get("SEMSEG_ARITH", "bf16x3")]
# which kernel runs the 16 batched row GEMMs of a Winograd forward / data gradient under bf16x3: "standalone" (256 x 128
# tiles, csrc/gemm_bf16split.hip; 197 vs 182 TFLOP/s fp32-equivalent on cls.0) or "igemm" (the SP instances of
# conv_igemm_kernel that the 1x1 convs run)
WINO_BF16X3_KERNEL = debug("wino_gemm", "standalone")
# SEMSEG_DEBUG=relu_bits=0: the fused BatchNorm-backward reductions read the post-ReLU activation as their mask (rounds 2-3) instead
# of the bit mask bn_apply writes next to it
RELU_BITS = debug("relu_bits", "1") != "0"
XCHG_HOST_OP = debug("xchg_host_op", "0") == "1"
WGRAD_EXACT_1X1_ONLY = debug("wgrad_exact_1x1_only", "0") == "1"     # measurement: the long-reduction rule for 1x1 convs only
WGRAD_BF16X3_MAX_M = int(debug("wgrad_bf16x3_max_m", "131072"))   # longer weight-gradient reductions: exact fp32 products
# Round 6, small per-GPU batch: a BatchNorm layer is four launches per pass pair (statistics -> finalize -> apply; reduction -> parameter
# gradients -> apply), the two middle ones ~5 us of pure latency each.  semseg_bn_apply_train / semseg_bn_bwd_apply_train derive scale /
# shift (the sums of g) inside the apply launch: every thread folds the slot replicas of its 4 channels itself.  Taken where the
# vector has ONE replica — behind a SyncBN all-reduce, i.e. on every layer of an N > 1 job (forced one-rank lines at per-GPU batch 2:
# 22.56 -> 22.10 ms peer exchange, 22.97 -> 22.41 ms RCCL) — and never at a large batch, where the flat grid of the separate apply
# kernel streams faster.  On one GPU the vectors keep their NSLOT replicas and the per-thread fold (nslot x 64 bytes of L2 reads per
# thread) costs what the launch saves: 20.68 ms separate, 20.78-20.85 ms fused up to 256-512 channels, 21.03 up to 1024
# (SEMSEG_DEBUG bn_fuse_max_cns = channels x replicas bound; 0 = one replica only).  A first form cut the replicas to 1-2 so that every
# layer qualified: the producers' same-address fp64 atomics then serialise, 21.7 -> 24.6 ms (profiles/r06_bs2_ab_notes.txt).
# SEMSEG_DEBUG=bn_fuse_small=0: four launches everywhere.
BN_FUSE_SMALL = debug("bn_fuse_small", "1") != "0"
BN_FUSE_MAX_M = int(debug("bn_fuse_max_m", "32768"))
BN_FUSE_MAX_CNS = int(debug("bn_fuse_max_cns", "0"))


def nslot_for(M):
    return ops.NSLOT


def bn_fuse_small(C, ns, M):
    return BN_FUSE_SMALL and C % 4 == 0 and M <= BN_FUSE_MAX_M and (ns == 1 or C * ns <= BN_FUSE_MAX_CNS)


def set_arith(name):
    """"bf16x3" | "f32" for engines built from now on; returns the previous name."""
    global ARITH
    old = arith_name()
    ARITH = _ARITH_NAMES[name]
    return old


def arith_name(a=None):
    a = ARITH if a is None else a
    return "f32" if a == ops.ARITH_F32 else "bf16x3"


def _fam(arith):
    """Suffix of a kernel-family label: the SP = 3 template instances run under bf16x3."""
    return ",SP3" if arith == ops.ARITH_BF16X3 else ""


class LabelWatch:
    """Ring of pinned host slots receiving, behind every training step's loss head, the number of labels that were neither
    ignore_index nor a class id (acc[2] of semseg_ce_head_fwd).  One per model.  Nothing is dropped: when the host runs
    more than RING steps ahead of the device, the oldest copy is waited for before its slot is reused."""
    RING = 8

    def __init__(self):
        self.host = torch.zeros(self.RING, dtype=F64).pin_memory()
        self.events = [None] * self.RING
        self.n = 0

    def watch(self, acc):
        i = self.n % self.RING
        self.n += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
            self._take(i, None)
        self.host[i:i + 1].copy_(acc[2:3], non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record()

    def _take(self, i, nclasses):
        self.events[i] = None
        nbad = int(self.host[i].item())
        if nbad:
            raise IndexError("Target out of bounds: %d label(s) of an earlier step were neither ignore_index nor in "
                             "[0, %s) (counted by the fused loss head)" % (nbad, "C" if nclasses is None else nclasses))

    def poll(self, nclasses=None, wait=False):
        for i, ev in enumerate(self.events):
            if ev is None:
                continue
            if wait:
                ev.synchronize()
            elif not ev.query():
                continue
            self._take(i, nclasses)


class ConvL:
    def __init__(self, mod, device, need_dgrad=True, training=True, name="", arith=ops.ARITH_F32):
        w = mod.weight
        self.mod = mod
        self.arith = arith      # per-launch argument of this layer's forward / data-gradient / weight-gradient GEMMs
        self.Co, self.Ci, self.R, self.S = w.shape
        self.stride, self.pad, self.dil = mod.stride[0], mod.padding[0], mod.dilation[0]
        # Winograd F(2x2, 3x3) for the stride-1 "same" 3x3 convs (csrc/winograd.hip): 1 / 2.25 of the
        # multiplications, the 16 GEMMs as one batched matrix-core launch.  Measured per shape at bs 16
        # (scripts/wino_bench.py, forward / data gradient / weight gradient, us): cls.0 15170 / 15240 / 17280 -> 8330 /
        # 8420 / 7130, layer4 conv2 1957 / 1948 / 2155 -> 1343 / 1295 / 1367, layer3 conv2 608 / 616 / 613 -> 427 / 393 /
        # 341; below 128 channels the transforms (HBM-bound) eat the gain (layer1 conv2: 152 -> 231), so those stay
        # direct.  Channel counts: K % 64 for the K-major GEMM, Co % 128 so that no padded column exists.
        # Eval engines use it too: the output transform carries the folded BatchNorm scale / shift, ReLU and residual.
        self.wino = None
        if (WINOGRAD and self.R == 3 and self.S == 3 and self.stride == 1 and self.pad == self.dil
                and self.Ci % 64 == 0 and self.Ci >= 128 and self.Co % 128 == 0 and mod.bias is None):
            self.wino = ops.WinoConv(self.Co, self.Ci, device, need_dgrad)
            self.pk = None
        else:
            self.pk = ops.PackedConv(self.Co, self.Ci, self.R, self.S, device, need_dgrad)
        self.wgrad = None
        self.bgrad = None


class BNL:
    def __init__(self, mod, eng):
        self.mod = mod
        self.C = mod.num_features
        self.eps = float(mod.eps)
        self.momentum = 0.1 if mod.momentum is None else float(mod.momentum)
        C = self.C
        self.stats = eng.alloc_f64(2 * C * ops.NSLOT)   # [NSLOT][2C]; slot 0 holds the combined vector
        self.sums = eng.alloc_f64(2 * C * ops.NSLOT)
        v = torch.empty(4 * C, dtype=F32, device=eng.device)
        self.mean, self.invstd, self.scale, self.shift = v[:C], v[C:2 * C], v[2 * C:3 * C], v[3 * C:]
        self.ggrad = None
        self.bgrad = None
        self.eval_epoch = -1
        self.ns = ops.NSLOT     # slot replicas of stats / sums the producers of THIS layer use (nslot_for(pixels), set by its producer)
        self.fin = None         # (stats vector, replicas, count, tracked): bn_prepare_group left the finalize step to bn_act's apply launch


# HIP streams (and the split-K scratch arenas that go with them) are shared by every engine of a process, per device.
# The runtime multiplexes streams onto a handful of hardware queues round-robin: when each engine created its own
# side / high-priority streams, the streams of the SECOND engine of a process (another input shape, an eval engine, the
# module path after the Trainer in bench.py) could land on one hardware queue, and its backward lost the concurrency
# of the weight-gradient stream with the dependent chain (measured: 207 ms for the first engine of a process, 221-230
# ms for an identical second one).  Engines of one process never run concurrently, so sharing is safe.
_SHARED = {}
MAX_SCRATCH_ARENAS = 4
# Bumped whenever a process-wide arena is replaced or released: a recorded step plan (semseg_amd/plan.py) holds the addresses
# of the arenas it saw and is discarded (re-recorded) when this moved since.
ARENA_GEN = [0]


def _shared(device, name, make):
    key = (device.index if device.index is not None else torch.cuda.current_device(), name)
    v = _SHARED.get(key)
    if v is None:
        v = _SHARED[key] = make()
    return v


class SyncGroup:
    """BatchNorm layers whose batch statistics do not depend on each other (bn3 + the downsample BN of a projection
    block, the four PPM branches, the cls / aux head BNs): their [2C] fp64 vectors sit side by side in ONE staging
    vector, so SyncBN needs one all-reduce per group and pass instead of one per layer (tool/train.py:142 converts
    every BatchNorm to nn.SyncBatchNorm, each of which does its own exchange).  `todo` / `finish` defer the backward
    of the members until the last one has delivered its sums."""

    def __init__(self, bls, device):
        self.bls = list(bls)
        self.buf = torch.zeros(sum(2 * b.C for b in bls), dtype=F64, device=device)
        self.views, off = {}, 0
        for b in bls:
            self.views[id(b)] = self.buf[off:off + 2 * b.C]
            off += 2 * b.C
        self.todo = 0
        self.finish = []

    def view(self, bl):
        return self.views[id(bl)]


class Engine:
    def __init__(self, model, N, H, W, training, kind):
        self.model = model
        self.kind = kind  # "psp" | "psa"
        self.N, self.H, self.W = N, H, W
        self.training = training
        self.device = next(model.parameters()).device
        assert self.device.type == "cuda", "the HIP engine needs the model on an MI355X (cuda) device"
        self._f64_chunks = []
        self._bufs = {}
        self._seq = 0
        self.tape = []
        self.convs = {}
        self.bns = {}
        self.arith = ARITH
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        # SEMSEG_FORCE_DIST=1 drives the N>1 code path (collectives included) on a 1-rank group: the
        # only way to exercise the RCCL calls on a single-GPU test box
        self.dist_on = self.world > 1 or (os.environ.get("SEMSEG_FORCE_DIST") == "1" and dist.is_initialized())
        self.sync_bn = self.dist_on and any(isinstance(m, nn.SyncBatchNorm) for m in model.modules())
        self.force_sync_bn = False
        self._f64_arena = None
        self._f64_off = 0
        self._f64_cap = 0
        self._register()
        self._flat_grads()
        self.grads_ready_hook = None  # callable(param_list) fired as parameter gradients complete
        self.weights_version = None
        self.ktimer = None
        self._eval_epoch = 0
        self.side_wgrad = debug("side_wgrad", "1") == "1"
        # Every weight gradient runs on the side stream (not only the small grids): the direct-to-LDS weight-gradient
        # kernel leaves >= 60 % of the VGPR file and 32 KB of LDS per CU free, so the HBM-bound BatchNorm backward
        # kernels and the data-gradient GEMMs of the main stream are co-resident with it instead of running alone
        # (measured bs 16: 217.4 -> 207.9 ms per step together with the high-priority chain below, 214.5 without
        # the side stream; scripts/step_variants.py).
        self.side_all = True
        # The dependent chain of backward (data gradients + BatchNorm) runs on a high-priority stream so that it wins
        # the dispatch race against the weight gradients queued on the side stream.
        # Not under torch.distributed: with the SyncBN all-reduces issued from the high-priority stream the forced
        # 1-rank RCCL step at batch 2 takes 60.5 ms instead of 39.2 (RCCL's own stream has normal priority and every
        # collective is an event round trip between the two).
        self.hipri_main = debug("hipri_main", "1") == "1" and not self.dist_on
        # number of weight-gradient streams used round-robin (each with its own split-K scratch); 2 / 3 / 4 measured
        # slower than 1
        self.n_side = 1
        # fold bn_bwd_reduce into the epilogue of the data gradient that completes a BatchNorm output's gradient
        self.fuse_bnr = debug("fuse_bnr", "1") == "1"
        self._sides, self._scr2s, self._side_rr = [], [], 0
        self._hi = None
        self._side = None
        self._side_used = False
        self._mod_ids = tuple(id(m) for m in model.modules())
        self._labels_checked = False
        self._drop_calls = 0
        # per-step part of the dropout call counter as the DEVICE sees it (semseg_dropout2d_mask's offset_dev): zero on the
        # eager path, advanced by Trainer before every replay of a recorded step (semseg_step_state_set)
        self.drop_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.recorder = None   # semseg_amd.plan.StepPlan while Trainer records a step of this engine
        # step-plan state, owned by Trainer (semseg_amd/trainer.py): the accepted record and its outputs, the candidate waiting for
        # the next step's record, the staging buffers the record points at, counters
        self._plan = None
        self._plan_candidate = None
        self._plan_out = None
        self._plan_x = self._plan_y = None
        self._plan_eager = 0       # launch-by-launch steps this engine has run under a plan-enabled Trainer
        self._plan_replays = 0
        self._plan_drops = 0       # dropout mask draws per step
        self._plan_drop_base = 0   # host dropout counter in front of the accepted record
        self._plan_sig = None      # Trainer._host_signature at record time
        self._plan_gen = -1        # ARENA_GEN at record time
        self._plan_tries = 0
        self._plan_off = False     # three recordings failed: the engine stays on the launch-by-launch path
        self.tape_hook = None  # callable(TapeOp) that must call op.fn(); set by tests only
        self._groups = {}
        self.syncbn_collectives_per_step = 0   # SyncBN all-reduces issued by the last forward + backward

    def push(self, kind, fn, **ctx):
        self.tape.append(TapeOp(kind, fn, ctx))

    def _run(self, op):
        if self.tape_hook is not None:
            self.tape_hook(op)
        else:
            op.fn()

    def params_stale(self):
        """True when the module tree changed under us (convert_sync_batchnorm, .to(device), ...)."""
        if tuple(id(m) for m in self.model.modules()) != self._mod_ids:
            return True
        p = self.params[0]
        return p.device != self.device

    # ------------------------------------------------------------------ registration
    def alloc_f64(self, n):
        if self._f64_arena is None:
            total = 0
            for m in self.model.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    total += 4 * m.num_features * ops.NSLOT
            total += 4096
            self._f64_arena = torch.zeros(total, dtype=F64, device=self.device)
            self._f64_cap = total
        assert self._f64_off + n <= self._f64_cap
        v = self._f64_arena[self._f64_off:self._f64_off + n]
        self._f64_off += n
        return v

    def _register(self):
        first = self.model.layer0[0]
        for name, m in self.model.named_modules():
            if isinstance(m, nn.Conv2d):
                if m is first:
                    self.convs[m] = None  # stem: direct kernel, no packed panel
                else:
                    # eval engines never run a data-gradient: no second packed panel (halves their weight copy)
                    self.convs[m] = ConvL(m, self.device, need_dgrad=self.training, training=self.training,
                                          name=name, arith=self.arith)
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                self.bns[m] = BNL(m, self)

    def _flat_grads(self):
        params = [p for p in self.model.parameters()]
        self.params = params
        total = sum(((p.numel() + 3) // 4) * 4 for p in params)
        self.flat_grad = torch.zeros(total, dtype=F32, device=self.device)
        self.grad_views = {}
        off = 0
        for p in params:
            n = p.numel()
            self.grad_views[p] = self.flat_grad[off:off + n].view(p.shape)
            off += ((n + 3) // 4) * 4
        for m, cl in self.convs.items():
            if cl is not None:
                cl.wgrad = self.grad_views[m.weight]
                cl.bgrad = self.grad_views[m.bias] if m.bias is not None else None
        for m, bl in self.bns.items():
            bl.ggrad = self.grad_views[m.weight]
            bl.bgrad = self.grad_views[m.bias]

    # ------------------------------------------------------------------ buffers
    def buf(self, shape, dtype=F32, zero=False, tag=""):
        key = (self._seq, tag)
        self._seq += 1
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        assert tuple(t.shape) == tuple(shape)
        return t

    def act(self, N, H, W, C, ld=None, zero=False, tag=""):
        ld = C if ld is None else ld
        t = self.buf((N, H, W, ld), zero=zero or ld != C, tag=tag)
        return Act(t, N, H, W, C, ld, tag)

    def grad_of(self, a):
        if a.grad is None:
            a.grad = self.buf((a.N, a.H, a.W, a.ld), zero=a.ld != a.C, tag="g:" + a.name)
        return a.grad

    # ------------------------------------------------------------------ weights
    def pack_weights(self):
        """One launch packs every conv weight (forward + data-gradient panels)."""
        import ctypes
        import numpy as np
        self._wino_filters()
        items = [(m, cl) for m, cl in self.convs.items() if cl is not None and cl.pk is not None]
        if not items:
            return
        ptrs = tuple(m.weight.data_ptr() for m, _ in items)
        if getattr(self, "_pack_ptrs", None) != ptrs:
            class Desc(ctypes.Structure):
                _fields_ = [("w", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_dgrad", ctypes.c_void_p),
                            ("Co", ctypes.c_int), ("Ci", ctypes.c_int), ("RS", ctypes.c_int),
                            ("Co_pad", ctypes.c_int), ("Ci_pad", ctypes.c_int), ("Kc_dgrad", ctypes.c_int)]
            arr = (Desc * len(items))()
            starts, blk = [], 0
            for i, (m, cl) in enumerate(items):
                pk = cl.pk
                RS = pk.R * pk.S
                arr[i] = Desc(m.weight.data_ptr(), pk.w_fwd.data_ptr(),
                              0 if pk.w_dgrad is None else pk.w_dgrad.data_ptr(), pk.Co, pk.Ci, RS,
                              pk.Co_pad, pk.Ci_pad, pk.Kc_dgrad)
                starts.append(blk)
                blk += (pk.Co_pad * pk.Ci * RS + 1023) // 1024
                starts.append(blk)   # a zero-length segment (no dgrad panel) is never selected by the block search
                if pk.w_dgrad is not None:
                    blk += (pk.Ci_pad * pk.Kc_dgrad * RS + 1023) // 1024
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self._pack_descs = torch.from_numpy(raw).to(self.device)
            self._pack_starts = torch.tensor(starts, dtype=torch.int32, device=self.device)
            self._pack_blocks = blk
            self._pack_n = len(items)
            self._pack_ptrs = ptrs
        ops.conv_pack_weights_multi(self._pack_descs, self._pack_starts, self._pack_n, self._pack_blocks)

    def _wino_filters(self):
        """Every Winograd filter panel (forward + flipped data-gradient panel per conv) in one launch."""
        import ctypes
        import numpy as np
        wl = [(m, cl.wino) for m, cl in self.convs.items() if cl is not None and cl.wino is not None]
        if not wl:
            return
        ptrs = tuple(m.weight.data_ptr() for m, _ in wl)
        if getattr(self, "_wf_ptrs", None) != ptrs:
            class Desc(ctypes.Structure):
                _fields_ = [("w", ctypes.c_void_p), ("U", ctypes.c_void_p), ("Co", ctypes.c_int), ("Ci", ctypes.c_int),
                            ("rows_pad", ctypes.c_int), ("Kc", ctypes.c_int), ("flip", ctypes.c_int)]
            panels = []
            for m, wc in wl:
                panels.append((m.weight.data_ptr(), wc.U_fwd.data_ptr(), wc.Co, wc.Ci, wc.Co_pad, wc.Ci, 0))
                if wc.U_dgrad is not None:
                    panels.append((m.weight.data_ptr(), wc.U_dgrad.data_ptr(), wc.Co, wc.Ci, wc.Ci_pad, wc.Kc, 1))
            arr = (Desc * len(panels))()
            starts, blk = [], 0
            for i, pn in enumerate(panels):
                arr[i] = Desc(*pn)
                starts.append(blk)
                blk += (pn[4] * pn[5] + 255) // 256
            raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self._wf_descs = torch.from_numpy(raw).to(self.device)
            self._wf_starts = torch.tensor(starts, dtype=torch.int32, device=self.device)
            self._wf_n, self._wf_blocks, self._wf_ptrs = len(panels), blk, ptrs
        ops.wino_filter_transform_multi(self._wf_descs, self._wf_starts, self._wf_n, self._wf_blocks)

    def _weights_sig(self):
        return tuple(p._version for p in self.params) + tuple(p.data_ptr() for p in self.params[:4])

    # ------------------------------------------------------------------ primitive layers
    def conv_bn(self, x, m, bm, relu=True, res=None, dropmask=None, out=None):
        """conv -> BatchNorm -> [ReLU].  Training: conv with fused statistics epilogue, then the BN apply
        kernel.  Eval: the BatchNorm (running statistics), the ReLU and an optional residual are folded into
        the conv epilogue (y = relu(acc*scale + shift + res)), so the raw conv output never reaches HBM."""
        if self.training:
            assert res is None
            return self.bn_act(self.conv(x, m, stats=self._st(bm)), bm, relu=relu, dropmask=dropmask, out=out)
        bl = self.bns[bm]
        self._eval_bn(bm, bl)
        return self.conv(x, m, out=out, fold=(bl.scale, bl.shift, relu, res))

    def _eval_bn(self, bm, bl):
        if bl.eval_epoch != self._eval_epoch:
            ops.bn_eval_params(bm.weight.detach(), bm.bias.detach(), bm.running_mean, bm.running_var, bl.eps,
                               bl.scale, bl.shift, bl.C)
            bl.eval_epoch = self._eval_epoch

    def conv(self, x, m, stats=None, out=None, bias=False, fold=None):
        cl = self.convs[m]
        Ho = ops.conv_out(x.H, cl.R, cl.stride, cl.pad, cl.dil)
        Wo = ops.conv_out(x.W, cl.S, cl.stride, cl.pad, cl.dil)
        assert x.C == cl.Ci, (x.C, cl.Ci)
        if out is None:
            ld = cl.Co if cl.Co % 64 == 0 else ops.roundup(cl.Co, 128)
            out = self.act(x.N, Ho, Wo, cl.Co, ld=ld, tag="conv")
        if cl.wino is not None:
            assert not bias
            T = ops.wino_tiles(x.N, x.H, x.W, cl.dil)
            if not self.training:
                # eval: the transformed input is scratch; BatchNorm (running statistics) / ReLU / residual ride in the
                # output transform exactly as they ride in the direct kernel's epilogue
                sc, sh, relu, res = fold if fold is not None else (None, None, False, None)
                self._wino_rows(x.data, x.ld, cl.Ci, cl.wino.U_fwd, cl.wino.Co_pad, out.data, out.ld, cl.Co, x.N, x.H,
                                x.W, cl.dil, T, self._wino_scratch("Vdy", 16 * T * cl.Ci), add=None if res is None else
                                res.data, ldadd=0 if res is None else res.ld, fold=(sc, sh, relu), arith=cl.arith)
                return out
            assert fold is None
            V = self.buf((16 * T * cl.Ci,), tag="winoV")      # kept: the weight gradient contracts it with dy
            if stats is not None:
                stats.ns = nslot_for(out.M)
            self._wino_rows(x.data, x.ld, cl.Ci, cl.wino.U_fwd, cl.wino.Co_pad, out.data, out.ld, cl.Co, x.N, x.H, x.W,
                            cl.dil, T, V, stats=stats, arith=cl.arith)
            if x.fuse_ok:
                x.pending += 1
            self.push("conv", lambda: self._conv_bwd_wino(x, out, cl, m, V, T), x=x, y=out, cl=cl, m=m)
            return out
        rs = cl.R * cl.S if cl.R * cl.S in (1, 9) else 0
        ar = cl.arith if rs else ops.ARITH_F32       # the generic tap walk has no split instance
        tile = ops.chosen_tile("fwd", cl.pk, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil, x.ld, out.ld, ar,
                               plain=fold is None and not (bias and m.bias is not None))
        fam = ("gemm_rows_bf16split_kernel<3,16,1%s> (bf16x3, 1x1 conv + statistics)" % (",128" if tile == ops.TILE_SPLIT_GEMM_128 else "")
               if tile in ops.SPLIT_GEMM_CODES else
               "conv_igemm_kernel<%d,%d,false,%d%s>(+splitk_epilogue)" % (64 if tile >= 1000 else 128, tile % 1000, rs, _fam(ar)))
        ev = self._t0(fam, 2.0 * x.N * Ho * Wo * cl.Co * cl.Ci * cl.R * cl.S)
        if fold is not None:
            sc, sh, relu, res = fold
            ops.conv_fwd(x.data, x.ld, cl.pk, out.data, out.ld, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil,
                         bias=sh, scale=sc, relu=relu, add=None if res is None else res.data,
                         ldadd=0 if res is None else res.ld, scratch=self.scratch(), arith=ar)
        else:
            if stats is not None:
                stats.ns = nslot_for(out.M)
            ops.conv_fwd(x.data, x.ld, cl.pk, out.data, out.ld, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil,
                         bias=m.bias.detach() if (bias and m.bias is not None) else None,
                         stats=None if stats is None else stats.stats,
                         nslot=1 if stats is None else stats.ns, scratch=self.scratch(), arith=ar)
        self._t1(ev)
        if self.training:
            if x.fuse_ok:
                x.pending += 1
            self.push("conv", lambda: self._conv_bwd(x, out, cl, m), x=x, y=out, cl=cl, m=m)
        return out

    def _wgrad(self, x, y, cl, m, scratch):
        dy = y.grad
        flops = 2.0 * y.M * cl.Co * cl.Ci * cl.R * cl.S
        big = cl.Ci % 128 == 0 and cl.Co >= 128
        # 128 x 128 tiles run the direct-to-LDS kernel (conv_wgrad.hip: WGRAD_DMA_POLICY; under bf16x3 its SP = 3 form,
        # WGRAD_SP_POLICY) and 64 x 64 tiles the register-staged fp32 kernel
        # Reductions longer than WGRAD_BF16X3_MAX_M pixels keep exact fp32 products: the three dropped cross products of
        # bf16x3 are an error of ~2^-24 PER PRODUCT, which a sum over M products with heavy cancellation carries as sqrt(M),
        # where the CPU's blocked fp32 sum grows much slower — in situ at the headline batch the 1x1 weight gradients at
        # 119 x 119 (M = 226 576: layer1.0.conv1, layer1.0.downsample.0, layer2.0.conv1) measured 3.6-4.2 x the CPU-fp32
        # recompute's rms error under bf16x3 (criterion 3 x) and 0.8-0.9 x with exact products (profiles/r05_insitu_b16.txt);
        # at M = 57 600 (every layer3 / layer4 / head conv of a batch-16 step) bf16x3 is inside the criterion.
        ar = cl.arith if (y.M <= WGRAD_BF16X3_MAX_M or (WGRAD_EXACT_1X1_ONLY and cl.R * cl.S > 1)) else ops.ARITH_F32
        ev = self._t0(self._wgrad_family(big, ar, cl.Ci), flops)
        ops.conv_wgrad(x.data, x.ld, dy, y.ld, cl.wgrad, scratch, x.N, x.H, x.W, cl.Ci,
                       cl.Co, cl.R, cl.S, cl.stride, cl.pad, cl.dil, arith=ar)
        self._t1(ev)
        ready = [m.weight]
        if m.bias is not None:
            # bias gradient = per-channel sum of dy (fp64 reduction, then the [C] cast kernel)
            C4 = ops.roundup(cl.Co, 4)
            st = self._bias_stats(C4)
            ops.zero_(st)
            ops.channel_stats(dy, y.ld, st, y.M, C4)
            ops.bn_param_grads(st, self._dummy(C4), cl.bgrad, cl.Co)
            ready.append(m.bias)
        self._ready(ready)

    @staticmethod
    def _wgrad_family(big, arith, Ci=0):
        if not big:
            return "conv_wgrad_kernel<64,64%s>+reduce" % _fam(arith)
        # conv_wgrad.hip: WGRAD_SP_POLICY 10 = the 128 x 256 kernel for layers with Ci % 256 == 0 (SEMSEG_DEBUG wgrad_sp overrides)
        if arith == ops.ARITH_BF16X3 and Ci % 256 == 0 and Ci > 0 and debug("wgrad_sp", "10") == "10":
            return "conv_wgrad_dma_wide_kernel<128x256,SP3>+reduce"
        return "conv_wgrad_dma_kernel<128x128%s>+reduce" % _fam(arith)

    def _conv_bwd(self, x, y, cl, m):
        dy = y.grad
        assert dy is not None
        flops = 2.0 * y.M * cl.Co * cl.Ci * cl.R * cl.S
        # Weight gradients are off the critical path of backward (nothing downstream reads them until
        # the optimizer / the gradient all-reduce).  When this conv's grid cannot fill the GPU on its
        # own (small per-GPU batch) it runs on a side HIP stream, concurrently with the data-gradient /
        # BatchNorm chain; both operands (x, dy) are final by now and stay untouched until the join at
        # the end of backward().
        side = self.side_wgrad and (self.side_all or y.M * cl.Co < 512 * 128 * 128)
        if side:
            st, scr = self._side_stream()
            ops.stream_wait(st, torch.cuda.current_stream())
            self._side_used = True      # before the block: _wgrad may hand a gradient bucket to the communicator
            with torch.cuda.stream(st):
                self._wgrad(x, y, cl, m, scr)
        else:
            self._wgrad(x, y, cl, m, self.scratch())
        if x.name != "input":
            gx = self.grad_of(x)
            # This data gradient completes x.grad when it is the last outstanding contribution; if x is a
            # BatchNorm(+ReLU) output whose consumers are all convs / residual adds, the BatchNorm-backward reduction
            # (mask + fp64 sum g, sum g*xhat) rides in its epilogue instead of a separate pass over HBM.
            last = x.fuse_ok and x.pending == 1
            if x.fuse_ok:
                x.pending -= 1
            bs = x.bnsrc
            fuse = (self.fuse_bnr and last and bs is not None and x.C % 4 == 0 and x.ld % 4 == 0 and
                    all(yk.ld % 4 == 0 for yk, _ in bs["bns"]))
            rs = cl.R * cl.S if cl.R * cl.S in (1, 9) else 0
            ar = cl.arith if rs else ops.ARITH_F32
            tile = ops.chosen_tile("dgrad", cl.pk, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil, y.ld, x.ld, ar,
                                   plain=not (fuse and len(bs["bns"]) > 1))
            fam = ("gemm_rows_bf16split_kernel<3,16,2%s> (bf16x3, 1x1 data gradient + fused reduction)"
                   % (",128" if tile == ops.TILE_SPLIT_GEMM_128 else "") if tile in ops.SPLIT_GEMM_CODES else
                   "conv_igemm_kernel<%d,%d,true,%d%s>(+splitk_epilogue)" % (64 if tile >= 1000 else 128, tile % 1000, rs, _fam(ar)))
            ev = self._t0(fam, flops)
            if fuse:
                ops.conv_dgrad_bnreduce(dy, y.ld, cl.pk, gx, x.ld, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil,
                                        x.data if bs["relu"] else None, x.ld,
                                        [(yk.data, yk.ld, blk.mean, blk.invstd, blk.sums) for yk, blk in bs["bns"]],
                                        bs["bns"][0][1].ns, add=gx if x.ginit else None, ldadd=x.ld, scratch=self.scratch(),
                                        arith=ar, relu_bits=x.bits if bs["relu"] else None)
                x.bn_reduced = True
            else:
                ops.conv_dgrad(dy, y.ld, cl.pk, gx, x.ld, x.N, x.H, x.W, cl.stride, cl.pad, cl.dil,
                               add=gx if x.ginit else None, ldadd=x.ld, scratch=self.scratch(), arith=ar)
            self._t1(ev)
            x.ginit = True

    def _wino_scratch(self, name, floats):
        """Scratch of the Winograd path (products M, transformed dy, dU), shared by the engines of the process and
        grown on demand; each name is used from one stream only (M / Vdy: the dependent chain, Yh / dU: the
        weight-gradient stream), so reuse is ordered."""
        key = (self.device.index if self.device.index is not None else torch.cuda.current_device(), "wino_" + name)
        t = _SHARED.get(key)
        if t is None or t.numel() < floats:
            if t is not None:
                torch.cuda.synchronize(self.device)     # a larger shape arrived: nothing may still read the old arena
                ARENA_GEN[0] += 1
            t = _SHARED[key] = (torch.zeros if name == "Yh" else torch.empty)(floats, dtype=F32, device=self.device)
        return t

    def _wino_rows(self, src, lds, K, U, rows_pad, dst, ldd, Nout, N, H, W, d, T, V, stats=None, add=None, ldadd=0,
                   bnr=None, fold=None, arith=ops.ARITH_F32):
        """input transform -> 16 batched row GEMMs [T x K] x [K x Nout] -> output transform: the forward of a Winograd conv
        (src = x, U = U_fwd) and its data gradient (src = dy, U = the flipped / transposed filter)."""
        px = N * H * W
        Mb = self._wino_scratch("M", 16 * T * Nout)
        ev = self._t0(WINO_HBM, -4.0 * (px * K + 16 * T * K))
        ops.wino_input_transform(src, lds, V, N, H, W, K, d)
        self._t1(ev)
        if arith == ops.ARITH_BF16X3 and WINO_BF16X3_KERNEL == "standalone" and K % 16 == 0 and rows_pad % 128 == 0:
            ev = self._t0("gemm_rows_bf16split_kernel<3,16> (bf16x3)", 2.0 * 16 * T * Nout * K)
            ops.gemm_rows_batched_bf16split(V, K, T * K, U, rows_pad * K, Mb, Nout, T * Nout, T, K, Nout, 16, nsplit=3)
        else:
            ev = self._t0("conv_igemm_kernel<128,%d,false,1%s>(+splitk_epilogue)" % (128 if Nout >= 128 else 64, _fam(arith)),
                          2.0 * 16 * T * Nout * K)
            ops.gemm_rows_batched(V, K, T * K, U, rows_pad * K, Mb, Nout, T * Nout, T, K, Nout, 16, arith=arith)
        self._t1(ev)
        ev = self._t0(WINO_HBM, -4.0 * (16 * T * Nout + px * Nout * (1 + (add is not None) + ((1 + (1.0 if bnr[7] is None else 1.0 / 32)) if bnr else 0))))
        if bnr is not None:
            act, ldact, ybn, ldybn, mean, invstd, sums, bits, ns = bnr
            ops.wino_output_transform_bnreduce(Mb, Nout, dst, ldd, N, H, W, Nout, d, act, ldact, ybn, ldybn, mean,
                                               invstd, sums, ns, add=add, ldadd=ldadd, relu_bits=bits)
        else:
            sc, sh, relu = fold if fold is not None else (None, None, False)
            ops.wino_output_transform(Mb, Nout, dst, ldd, N, H, W, Nout, d, add=add, ldadd=ldadd,
                                      stats=None if stats is None else stats.stats, nslot=1 if stats is None else stats.ns,
                                      scale=sc, shift=sh, relu=relu)
        self._t1(ev)

    def _conv_bwd_wino(self, x, y, cl, m, V, T):
        """Backward of a Winograd conv: weight gradient dU[e] = Yh[e]^T V[e] on the side stream (one batched K-major GEMM,
        V kept from forward), data gradient = the forward machinery on dy with the flipped filter on the main chain."""
        dy = y.grad
        assert dy is not None
        N, H, W, d = x.N, x.H, x.W, cl.dil
        gflops = 2.0 * 16 * T * cl.Co * cl.Ci

        def wgrad(scr):
            Yh, dU = self._wino_scratch("Yh", 16 * T * cl.Co), self._wino_scratch("dU", 16 * cl.Co * cl.Ci)
            px = N * H * W
            ev = self._t0(WINO_HBM, -4.0 * (px * cl.Co + 16 * T * cl.Co))
            ops.wino_dy_transform_wgrad(dy, y.ld, Yh, cl.Co, N, H, W, cl.Co, d)
            self._t1(ev)
            ev = self._t0(self._wgrad_family(True, cl.arith, cl.Ci), gflops)
            ops.gemm_kmajor_batched(V, cl.Ci, T * cl.Ci, Yh, cl.Co, T * cl.Co, dU, cl.Co * cl.Ci, scr, T, cl.Ci, cl.Co, 16,
                                    arith=cl.arith)
            self._t1(ev)
            ev = self._t0(WINO_HBM, -4.0 * 25 * cl.Co * cl.Ci)
            ops.wino_filter_grad(dU, cl.wgrad, cl.Co, cl.Ci)
            self._t1(ev)
            self._ready([m.weight])

        if self.side_wgrad:
            st, scr = self._side_stream()
            ops.stream_wait(st, torch.cuda.current_stream())
            self._side_used = True
            with torch.cuda.stream(st):
                wgrad(scr)
        else:
            wgrad(self.scratch())
        if x.name != "input":
            gx = self.grad_of(x)
            # as in _conv_bwd: when this data gradient completes x.grad and x is the output of ONE BatchNorm(+ReLU) whose
            # consumers are all convs / residual adds, that layer's backward reduction rides in the output transform
            last = x.fuse_ok and x.pending == 1
            if x.fuse_ok:
                x.pending -= 1
            bs = x.bnsrc
            bnr = None
            if (self.fuse_bnr and last and bs is not None and len(bs["bns"]) == 1 and x.C % 4 == 0 and x.ld % 4 == 0
                    and bs["bns"][0][0].ld % 4 == 0):
                yk, blk = bs["bns"][0]
                bnr = (x.data if bs["relu"] else None, x.ld, yk.data, yk.ld, blk.mean, blk.invstd, blk.sums,
                       x.bits if bs["relu"] else None, blk.ns)
            self._wino_rows(dy, y.ld, cl.wino.Kc, cl.wino.U_dgrad, cl.wino.Ci_pad, gx, x.ld, cl.Ci, N, H, W, d, T,
                            self._wino_scratch("Vdy", 16 * T * cl.wino.Kc), add=gx if x.ginit else None, ldadd=x.ld,
                            bnr=bnr, arith=cl.arith)
            if bnr is not None:
                x.bn_reduced = True
            x.ginit = True

    def _side_stream(self):
        """Next weight-gradient stream (round-robin over n_side) and its private split-K scratch."""
        while len(self._sides) < max(1, self.n_side):
            i = len(self._sides)
            self._sides.append(_shared(self.device, "side%d" % i, lambda: torch.cuda.Stream(device=self.device)))
            self._scr2s.append(_shared(self.device, "side_scratch%d" % i,
                                       lambda: torch.empty(64 * 1024 * 1024, dtype=F32, device=self.device)))
        i = self._side_rr % max(1, self.n_side)
        self._side_rr += 1
        self._side = self._sides[0]
        return self._sides[i], self._scr2s[i]

    def scratch(self):
        """256 MB arena for split-K partial slabs (conv fwd/dgrad at small batch, every wgrad), one per stream that
        work is issued on (shared by the engines of the process: work on one stream is ordered)."""
        st = torch.cuda.current_stream(self.device)
        dev = self.device.index if self.device.index is not None else torch.cuda.current_device()
        lru = _SHARED.setdefault((dev, "scratch_lru"), [])      # [(stream handle, arena)], most recently used last
        for i, (h, t) in enumerate(lru):
            if h == st.cuda_stream:
                if i != len(lru) - 1:
                    lru.append(lru.pop(i))
                return t
        # at most MAX_SCRATCH_ARENAS (256 MB each) are kept: the streams the engines themselves use (the caller's, the
        # high-priority chain) plus a few caller streams; the least recently used one is released after a device
        # synchronisation (nothing can still be reading it)
        if len(lru) >= MAX_SCRATCH_ARENAS:
            torch.cuda.synchronize(self.device)
            lru.pop(0)
            ARENA_GEN[0] += 1
        t = torch.empty(64 * 1024 * 1024, dtype=F32, device=self.device)
        lru.append((st.cuda_stream, t))
        return t

    def _t0(self, family, flops):
        if self.ktimer is None:
            return None
        s, e = self.ktimer.span(family, flops)
        s.record()
        return e

    def _t1(self, ev):
        if ev is not None:
            ev.record()

    def _bias_stats(self, C4):
        if not hasattr(self, "_bstats"):
            self._bstats = self.alloc_f64(2 * 1024)
            self._bdummy = torch.empty(1024, dtype=F32, device=self.device)
        return self._bstats[:2 * C4]

    def _dummy(self, C4):
        return self._bdummy

    def _ready(self, plist):
        if self.grads_ready_hook is not None:
            self.grads_ready_hook(plist)

    def order_after_all_producers(self):
        """Gradients are written by two streams (the main one and the weight-gradient side stream), and a
        collective is only ordered after the stream that is current when it is issued.  Make that stream
        wait for everything the other one has enqueued so far; call this right before a gradient bucket
        is handed to the communicator (the call may come from either stream)."""
        if not self._sides or not self._side_used:
            return
        cur = torch.cuda.current_stream()
        for st in self._sides + [self._main]:
            if st != cur:
                ops.stream_wait(cur, st)

    def _syncing(self):
        return (self.sync_bn or self.force_sync_bn) and self.dist_on

    def _all_reduce(self, t, src=None, nslot=1):
        """One SyncBN exchange: the peer-memory exchange kernel (semseg_amd/syncbn_xchg.py: one launch, no c10d call) when
        it passed its start-up self-test among the job's ranks or is forced, RCCL otherwise.  src / nslot: the vector still
        lies in `nslot` slot replicas (src = [nslot][len(t)], t = its first slot): the exchange kernel folds them itself, the
        RCCL path needs them folded first (semseg_bn_combine) — one launch less per forward BatchNorm layer on the former."""
        from . import syncbn_xchg
        xc = syncbn_xchg.active(self.device)
        if src is not None and xc is None:
            ops.bn_combine(src, nslot, t.numel() // 2)
            src = None

        if xc is not None:
            # one launch on the compute stream, exchange number in device memory: an entry of the recorded step like any other
            # (SEMSEG_XCHG_HOST_OP=1: issued as a host operation between two C segments instead, the form before the counter
            # moved into device memory; A/B only)
            def exchange():
                if src is not None:
                    xc.all_reduce(src, nslot=nslot, n=t.numel(), out=t)
                else:
                    xc.all_reduce(t)
            if XCHG_HOST_OP:
                self.host_op(exchange)
            else:
                exchange()
        else:
            self.host_op(lambda: dist.all_reduce(t))
        self.syncbn_collectives_per_step += 1

    def host_op(self, fn):
        """Runs a host-side operation of the step (a torch.distributed collective).  While a step plan records this engine's
        launches (semseg_amd/plan.py) the operation also becomes a segment boundary of the plan and is re-issued, on the stream
        that is current now, by every replay."""
        if self.recorder is None:
            fn()
        else:
            self.recorder.py_op(fn)

    def _group(self, bls):
        key = tuple(id(b) for b in bls)
        g = self._groups.get(key)
        if g is None:
            g = self._groups[key] = SyncGroup(bls, self.device)
        return g

    def bn_prepare(self, bm, count):
        """stats -> scale/shift (train: batch statistics, SyncBN all-reduce; eval: running stats)."""
        return self.bn_prepare_group([(bm, count)])[0]

    def bn_prepare_group(self, items, fuse_ok=True):
        """items: [(BatchNorm module, values per channel on this rank)] of layers whose statistics are all complete.
        Under SyncBN the group's [sum, sum of squares] vectors are combined into one staging vector and all-reduced
        ONCE.  Returns the global count per layer (0 for layers normalising with running statistics).
        fuse_ok: the caller's apply launch handles ONE BatchNorm, so for layers with <= 2 slot replicas the finalize step is
        left to it (BNL.fin -> semseg_bn_apply_train) instead of being launched here."""
        out = []
        train = [(bm, c) for bm, c in items if self.training and bm.training]
        if train and self._syncing():
            bls = [self.bns[bm] for bm, _ in train]
            if len(bls) == 1:
                bl = bls[0]
                views = [bl.stats[:2 * bl.C]]
                if bl.ns == 1:
                    self._all_reduce(views[0])          # one replica: the vector is what SyncBN all-reduces as it lies
                else:
                    self._all_reduce(views[0], src=bl.stats, nslot=bl.ns)
            else:
                g = self._group(bls)
                views = [g.view(bl) for bl in bls]
                for bl, v in zip(bls, views):
                    ops.bn_combine(bl.stats, bl.ns, bl.C, dst=v)
                self._all_reduce(g.buf)
            src = {id(bm): (v, 1) for (bm, _), v in zip(train, views)}
        else:
            src = {id(bm): (self.bns[bm].stats, self.bns[bm].ns) for bm, _ in train}
        for bm, count in items:
            bl = self.bns[bm]
            bl.fin = None
            if id(bm) in src:
                st, ns = src[id(bm)]
                cnt = count * self.world if self._syncing() else count
                if cnt <= 1:
                    raise ValueError("Expected more than 1 value per channel when training, got input "
                                     "size [%d values per channel]" % cnt)
                track = bm.track_running_stats and bm.running_mean is not None
                if fuse_ok and bn_fuse_small(bl.C, ns, count):
                    bl.fin = (st, ns, cnt, track)
                    out.append(cnt)
                    continue
                # (folding this step into the apply kernel for EVERY layer — every thread deriving scale / shift of its 4
                # channels from the [nslot][2C] sums — was built and measured in round 4: the ~1 M threads of an apply launch
                # re-read 512 B of sums each, +8 ms per step at batch 16; DESIGN.md 8.5.  Round 6 does it where there are <= 2
                # replicas, with a bounded grid: bl.fin above)
                ops.bn_finalize(st, cnt, bm.weight.detach(), bm.bias.detach(),
                                bm.running_mean if track else None, bm.running_var if track else None,
                                bm.num_batches_tracked if track else None, bl.momentum, bl.eps, bl.mean,
                                bl.invstd, bl.scale, bl.shift, bl.C, nslot=ns)
                out.append(cnt)
            else:
                ops.bn_eval_params(bm.weight.detach(), bm.bias.detach(), bm.running_mean, bm.running_var, bl.eps,
                                   bl.scale, bl.shift, bl.C)
                out.append(0)
        return out

    def bn_act(self, y, bm, relu=True, res=None, y2=None, bm2=None, dropmask=None, out=None, prepared=None,
               group=None):
        """out = [relu](bn(y) (+ bn2(y2)) (+ res)) (* dropmask)   — model/resnet.py:76-92.
        prepared / group: the caller already ran bn_prepare_group for this layer together with others (PPM branches,
        the two heads); `group` defers this layer's backward until the group's one all-reduce."""
        bl = self.bns[bm]
        bl2 = None
        if y2 is not None:
            assert prepared is None
            bl2 = self.bns[bm2]
            cnt = self.bn_prepare_group([(bm, y.M), (bm2, y2.M)], fuse_ok=False)[0]
        else:
            cnt = self.bn_prepare(bm, y.M) if prepared is None else prepared
        if out is None:
            out = self.act(y.N, y.H, y.W, y.C, tag="bnact")
        # The ReLU mask as bits: what the fused BatchNorm-backward reductions read in backward instead of the activation
        # itself (1/32 of its bytes; the activation is the largest operand of a 1x1 data gradient's epilogue).
        if RELU_BITS and self.training and relu and dropmask is None and y.C % 32 == 0:
            out.bits = self.buf((y.M, y.C // 32), dtype=torch.int32, tag="relubits")
        if bl.fin is not None:
            assert y2 is None
            st, ns, fcnt, track = bl.fin
            bl.fin = None
            ops.bn_apply_train(y.data, y.ld, st, ns, fcnt, bm.weight.detach(), bm.bias.detach(),
                               bm.running_mean if track else None, bm.running_var if track else None,
                               bm.num_batches_tracked if track else None, bl.momentum, bl.eps, bl.mean, bl.invstd,
                               out.data, out.ld, y.M, y.C, y.H * y.W, relu, res=None if res is None else res.data,
                               ldres=0 if res is None else res.ld, dropmask=dropmask, relu_bits=out.bits)
        else:
            ops.bn_apply(y.data, y.ld, bl.scale, bl.shift, out.data, out.ld, y.M, y.C, y.H * y.W, relu,
                         y2=None if y2 is None else y2.data, ldy2=0 if y2 is None else y2.ld,
                         scale2=None if bl2 is None else bl2.scale, shift2=None if bl2 is None else bl2.shift,
                         res=None if res is None else res.data, ldres=0 if res is None else res.ld,
                         dropmask=dropmask, relu_bits=out.bits)
        if self.training:
            if dropmask is None:
                out.bnsrc = dict(bns=[(y, bl)] + ([(y2, bl2)] if y2 is not None else []), relu=relu)
            if res is not None and res.fuse_ok:
                res.pending += 1
            if group is not None:
                group.todo += 1
            self.push("bn_act", lambda: self._bn_act_bwd(y, bm, bl, relu, res, y2, bm2, bl2, dropmask, out, cnt, group),
                      y=y, bm=bm, bl=bl, relu=relu, res=res, y2=y2, bm2=bm2, bl2=bl2, dropmask=dropmask, out=out,
                      cnt=cnt)
        return out

    def _bn_act_bwd(self, y, bm, bl, relu, res, y2, bm2, bl2, dropmask, out, cnt, group=None):
        dout = out.grad
        assert dout is not None and out.ginit
        gy = self.grad_of(y)
        if res is not None and res.fuse_ok:
            res.pending -= 1
        if out.bn_reduced:
            # the data gradient that completed out.grad already masked it and accumulated the sums of this layer
            # (and of the downsample BN): out.grad IS g
            g, ldg = dout, out.ld
            if res is not None:
                assert not res.ginit and res.grad is None and res.ld == out.ld
                res.grad = dout           # the residual's gradient starts as g; later contributions add in place
                res.ginit = True
        else:
            if res is not None:
                assert not res.ginit
                g, ldg = self.grad_of(res), res.ld
                res.ginit = True
            else:
                g, ldg = gy, y.ld
            ops.bn_bwd_reduce(dout, out.ld, out.data if relu else None, out.ld, dropmask, y.H * y.W, y.data,
                              y.ld, bl.mean, bl.invstd, g, ldg, bl.sums, y.M, y.C, nslot=bl.ns)
            if y2 is not None:
                ops.bn_bwd_reduce(g, ldg, None, 0, None, y2.H * y2.W, y2.data, y2.ld, bl2.mean, bl2.invstd,
                                  None, 0, bl2.sums, y2.M, y2.C, nslot=bl2.ns)
        members = ([(y2, bm2, bl2)] if y2 is not None else []) + [(y, bm, bl)]
        sync = self._syncing()
        if sync and group is None and len(members) > 1:
            group = self._group([bl, bl2])    # bn3 + downsample BN: a group that is complete within this op
            group.todo = 1
        if not sync:
            group = None
        if group is None and len(members) == 1 and bn_fuse_small(bl.C, 1 if sync else bl.ns, y.M):
            # small tensor, one BatchNorm, no group: the parameter gradients ride in the backward apply launch (and the slot
            # replicas are folded there).  Under SyncBN it follows the all-reduce: the sums are global and every rank writes
            # global / world — what the gradient all-reduce + 1 / world makes of torch's per-rank local gradients too.
            if sync:
                self._all_reduce(bl.sums[:2 * bl.C], src=bl.sums if bl.ns > 1 else None, nslot=bl.ns)
            ops.bn_bwd_apply_train(g, ldg, y.data, y.ld, bl.mean, bl.invstd, bm.weight.detach(), bl.sums,
                                   1 if sync else bl.ns, cnt, 1.0 / self.world if sync else 1.0, bl.ggrad, bl.bgrad,
                                   self.grad_of(y), y.ld, y.M, y.C)
            y.ginit = True
            self._ready([bm.weight, bm.bias])
            return
        # parameter gradients come from the LOCAL sums (torch SyncBatchNorm semantics); the folded [2C] vector — what
        # the input gradient needs summed over all ranks — lands in slot 0 or in the group's staging piece
        for yy, bmm, bll in members:
            ops.bn_param_grads(bll.sums, bll.ggrad, bll.bgrad, bll.C, nslot=bll.ns,
                               folded=None if group is None else group.view(bll))

        def finish():
            for yy, bmm, bll in members:
                ops.bn_bwd_apply(g, ldg, yy.data, yy.ld, bll.mean, bll.invstd, bmm.weight.detach(),
                                 bll.sums if group is None else group.view(bll), cnt, self.grad_of(yy), yy.ld, yy.M,
                                 yy.C)
                yy.ginit = True
                self._ready([bmm.weight, bmm.bias])

        if group is None:
            if sync:
                self._all_reduce(bl.sums[:2 * bl.C])
            finish()
        else:
            group.finish.append(finish)
            group.todo -= 1
            if group.todo == 0:
                self._all_reduce(group.buf)
                fs, group.finish = group.finish, []
                for f in fs:
                    f()

    # ------------------------------------------------------------------ network pieces
    def stem(self, x_nchw):
        """layer0 = conv-bn-relu x3 + maxpool (model/resnet.py:106-115, model/pspnet.py:46)."""
        l0 = self.model.layer0
        N, _, H, W = x_nchw.shape
        c0 = l0[0]
        Ho, Wo = ops.conv_out(H, 3, 2, 1, 1), ops.conv_out(W, 3, 2, 1, 1)
        y0 = self.act(N, Ho, Wo, 64, tag="stem0")
        w0 = c0.weight.detach()
        ops.stem_conv_fwd(x_nchw, w0, y0.data, N, H, W)
        bl = self.bns[l0[1]]
        if self.training and l0[1].training:
            bl.ns = nslot_for(y0.M)
            ops.channel_stats(y0.data, y0.ld, bl.stats, y0.M, 64, nslot=bl.ns)
        if self.training:
            def bwd():
                ops.stem_conv_wgrad(x_nchw, y0.grad, self.grad_views[c0.weight], N, H, W, scratch=self.scratch())
                self._ready([c0.weight])
            self.push("stem_wgrad", bwd, x=x_nchw, y=y0, m=c0)
        a = self.bn_act(y0, l0[1])
        a = self.conv_bn(a, l0[3], l0[4])
        a = self.conv_bn(a, l0[6], l0[7])
        # maxpool
        Hp, Wp = ops.conv_out(a.H, 3, 2, 1, 1), ops.conv_out(a.W, 3, 2, 1, 1)
        p = self.act(N, Hp, Wp, a.C, tag="pool")
        idx = self.buf((N, Hp, Wp, a.C // 4), dtype=torch.int32, tag="poolidx")
        ops.maxpool_fwd(a.data, p.data, idx, N, a.H, a.W, a.C)
        if self.training:
            def bwd_pool():
                ga = self.grad_of(a)
                ops.maxpool_bwd(p.grad, idx, ga, N, a.H, a.W, a.C)
                a.ginit = True
            self.push("maxpool", bwd_pool, x=a, y=p, idx=idx)
        return p

    def _st(self, bm):
        """The BNL whose statistics the producing conv accumulates (None: the layer normalises with running statistics)."""
        return self.bns[bm] if (self.training and bm.training) else None

    def bottleneck(self, x, blk, out=None):
        a1 = self.conv_bn(x, blk.conv1, blk.bn1)
        a1.fuse_ok = self.training            # consumed by conv2 only
        a2 = self.conv_bn(a1, blk.conv2, blk.bn2)
        a2.fuse_ok = self.training            # consumed by conv3 only
        if not self.training:
            r = x if blk.downsample is None else self.conv_bn(x, blk.downsample[0], blk.downsample[1], relu=False)
            return self.conv_bn(a2, blk.conv3, blk.bn3, relu=True, res=r, out=out)
        y3 = self.conv(a2, blk.conv3, stats=self._st(blk.bn3))
        if blk.downsample is not None:
            yd = self.conv(x, blk.downsample[0], stats=self._st(blk.downsample[1]))
            o = self.bn_act(y3, blk.bn3, y2=yd, bm2=blk.downsample[1], out=out)
        else:
            o = self.bn_act(y3, blk.bn3, res=x, out=out)
        # A block output is consumed by the next block's conv1 (+ downsample conv) and as its residual, or by the head
        # convs: all tracked by Act.pending.  Not when it is a channel slice of the head's concat buffer (the PPM / PSA
        # pooling and the concat-wide cls conv read it in ways the counter does not see).
        o.fuse_ok = out is None
        return o

    def trunk(self, x_nchw, cat_C):
        """layer0..layer4; layer4's output lands in channels [0,2048) of the head's concat buffer."""
        m = self.model
        a = self.stem(x_nchw)
        for blk in m.layer1:
            a = self.bottleneck(a, blk)
        for blk in m.layer2:
            a = self.bottleneck(a, blk)
        for blk in m.layer3:
            a = self.bottleneck(a, blk)
        x_tmp = a
        blocks = list(m.layer4)
        cat = None
        for i, blk in enumerate(blocks):
            if i == len(blocks) - 1 and cat_C > 2048:
                cat = self.act(a.N, a.H, a.W, cat_C, tag="cat")
                a = self.bottleneck(a, blk, out=cat.slice(0, 2048))
            else:
                a = self.bottleneck(a, blk)
        return x_tmp, a, cat

    def ppm(self, x4, cat):
        """PPM (model/pspnet.py:8-26): pooled 1x1 conv-bn-relu branches upsampled into the concat."""
        feats = self.model.ppm.features
        bins = []
        for f in feats:
            b = f[0].output_size
            bins.append(b if isinstance(b, int) else b[0])
        bins = tuple(bins)
        N, H, W, C = x4.N, x4.H, x4.W, x4.C
        tot = sum(N * b * b * C for b in bins)
        pooled = self.buf((tot,), tag="pooled")
        ops.adaptive_avgpool_fwd(x4.data, x4.ld, pooled, bins, N, H, W, C, scratch=self.scratch())
        off = 0
        pacts = []
        c0 = C
        for f, b in zip(feats, bins):
            n = N * b * b * C
            pacts.append(Act(pooled[off:off + n].view(N, b, b, C), N, b, b, C, C, "pooled%d" % b))
            off += n
        ys = cnts = grp = None
        if self.training:
            # the four branch convs first: their BatchNorm statistics do not depend on each other, so under SyncBN
            # they travel in ONE all-reduce per pass (bn_prepare_group) instead of four
            ys = [self.conv(pa, f[1], stats=self._st(f[2])) for pa, f in zip(pacts, feats)]
            cnts = self.bn_prepare_group([(f[2], yb.M) for f, yb in zip(feats, ys)])
            if self._syncing() and all(f[2].training for f in feats):
                grp = self._group([self.bns[f[2]] for f in feats])
        for i, (f, b) in enumerate(zip(feats, bins)):
            pa = pacts[i]
            if self.training:
                ab = self.bn_act(ys[i], f[2], prepared=cnts[i], group=grp)
            else:
                ab = self.conv_bn(pa, f[1], f[2])
            dst = cat.slice(c0, ab.C)
            ops.bilinear_fwd(ab.data, ab.ld, dst.data, dst.ld, N, b, b, H, W, ab.C)
            if self.training:
                def bwd_up(ab=ab, c0=c0, b=b):
                    gab = self.grad_of(ab)
                    ops.bilinear_bwd(cat.grad[..., c0:], cat.ld, gab, ab.ld, N, b, b, H, W, ab.C)
                    ab.ginit = True
                self.push("upsample", bwd_up, x=ab, dy=lambda c0=c0: cat.grad[..., c0:], lddy=cat.ld, Ho=H, Wo=W)
            c0 += ab.C
        if self.training:
            dpool = self.buf((tot,), tag="dpool")
            o = 0
            for pa, b in zip(pacts, bins):
                n = N * b * b * C
                pa.grad = dpool[o:o + n].view(N, b, b, C)
                o += n

            def bwd_pool():
                # x4.grad aliases cat.grad[..., :C]; the pooled-branch gradients are added in place
                gx = cat.grad
                ops.adaptive_avgpool_bwd(gx, cat.ld, dpool, gx, cat.ld, bins, N, H, W, C)
                x4.grad = gx
                x4.ginit = True
            # must run after every branch backward => insert *before* the branch ops on the tape
            self._ppm_pool_bwd = TapeOp("ppm_pool", bwd_pool, dict(cat=cat, dpool=dpool, bins=bins, C=C))
        return cat

    def head(self, x, seq, tag):
        """cls / aux: conv3x3-bn-relu-dropout2d-conv1x1(+bias) (model/pspnet.py:64-78)."""
        conv_a, bn_a, drop, conv_b = seq[0], seq[1], seq[3], seq[4]
        dm = None
        if self.training and drop.training and drop.p > 0:
            dm = self.buf((x.N, conv_a.weight.shape[0]), tag="dropmask")
            # seeded from torch's generator state (torch.manual_seed reproduces a run), advanced per call
            self._drop_calls += 1
            ops.dropout2d_mask(dm, drop.p, torch.initial_seed(), self._drop_calls, self.drop_dev)
        a = self.conv_bn(x, conv_a, bn_a, dropmask=dm)
        ncls = conv_b.weight.shape[0]
        out = self.act(x.N, x.H, x.W, ncls, ld=ops.roundup(ncls, 128), tag="scores" + tag)
        return self.conv(a, conv_b, out=out, bias=True)

    def heads_train(self, feat, x_tmp):
        """cls and aux heads of a training step (model/pspnet.py:96-103).  Same layers as head(), but both 3x3 convs
        run before either BatchNorm: the two BatchNorms then share one SyncBN all-reduce per pass."""
        hs = []
        for x, seq in ((feat, self.model.cls), (x_tmp, self.model.aux)):
            conv_a, bn_a, drop = seq[0], seq[1], seq[3]
            dm = None
            if drop.training and drop.p > 0:
                dm = self.buf((x.N, conv_a.weight.shape[0]), tag="dropmask")
                self._drop_calls += 1
                ops.dropout2d_mask(dm, drop.p, torch.initial_seed(), self._drop_calls, self.drop_dev)
            hs.append((self.conv(x, conv_a, stats=self._st(bn_a)), dm))
        bns = [self.model.cls[1], self.model.aux[1]]
        cnts = self.bn_prepare_group([(bm, yh.M) for bm, (yh, _) in zip(bns, hs)])
        grp = self._group([self.bns[bm] for bm in bns]) if (self._syncing() and all(b.training for b in bns)) else None
        outs = []
        for seq, bm, (yh, dm), cnt, tag in zip((self.model.cls, self.model.aux), bns, hs, cnts, ("m", "a")):
            a = self.bn_act(yh, bm, dropmask=dm, prepared=cnt, group=grp)
            ncls = seq[4].weight.shape[0]
            out = self.act(yh.N, yh.H, yh.W, ncls, ld=ops.roundup(ncls, 128), tag="scores" + tag)
            outs.append(self.conv(a, seq[4], out=out, bias=True))
        return outs

    def ce(self, scores, label, H, W, ignore_index, want_pred, tag):
        N = scores.N
        lse = self.buf((N, H, W), tag="lse" + tag)
        pred = self.buf((N, H, W), dtype=torch.int64, tag="pred" + tag) if want_pred else None
        acc = self.buf((3,), dtype=F64, tag="acc" + tag)
        # both losses sit in ONE [2] buffer: what the callers see are slices of a clone of it (one launch per step, Trainer.step /
        # module_base._NetFunction), so a loss tensor kept across steps keeps its value — the reference returns fresh tensors
        # (model/pspnet.py:101-103)
        if tag == "m":
            self._losses = self.buf((2,), tag="losses")
        loss = self._losses[0:1] if tag == "m" else self._losses[1:2]
        ops.ce_head_fwd(scores.data, scores.ld, label, lse, pred, acc, loss, N, scores.H, scores.W, H, W,
                        scores.C, ignore_index)
        rec = dict(scores=scores, label=label, lse=lse, acc=acc, H=H, W=W, ignore=ignore_index)
        return loss, pred, rec

    # Out-of-range class ids: torch's CrossEntropyLoss (the reference's criterion, tool/train.py:121) raises on them; the
    # fused head counts them (acc[2]) and treats them as ignored.  The first step of an engine checks synchronously
    # (label_check, below); EVERY later step copies its count into a ring of pinned host slots behind the head kernel
    # (LabelWatch, one per model, shared by its engines), and the next forward / Trainer.step / eval forward / explicit
    # check_labels() that finds copies complete raises — no synchronisation on the training path, no step skipped.
    def _label_watch(self):
        w = self.model.__dict__.get("_hip_label_watch")
        if w is None:
            w = self.model.__dict__["_hip_label_watch"] = LabelWatch()
        return w

    def check_labels(self):
        """Blocks until the label counts of all watched steps are on the host and raises if any is non-zero."""
        self._label_watch().poll(self.model.cls[4].weight.shape[0], wait=True)

    def ce_bwd(self, rec, gloss):
        s = rec["scores"]
        g = self.grad_of(s)
        ops.ce_head_bwd(s.data, s.ld, rec["label"], rec["lse"], rec["acc"], gloss, 1.0, g, s.ld, False,
                        s.N, s.H, s.W, rec["H"], rec["W"], s.C, rec["ignore"], scratch=self.scratch())
        s.ginit = True

    # ------------------------------------------------------------------ whole-network passes
    def out_hw(self):
        z = self.model.zoom_factor
        return int((self.H - 1) / 8 * z + 1), int((self.W - 1) / 8 * z + 1)

    def _begin(self, x):
        assert x.is_cuda and x.dtype == F32 and tuple(x.shape) == (self.N, 3, self.H, self.W)
        self._seq = 0
        self.tape = []
        # parameter version counters + the model-wide counter of HIP training steps (the fused SGD
        # kernel updates weights without touching torch's version counters)
        sig = (self._weights_sig(), 0 if self.training else self.model.__dict__.get("_hip_bn_epoch", 0))
        if self.training or sig != self.weights_version:
            self.pack_weights()
            self.weights_version = sig
        if self.training:
            ops.zero_(self._f64_arena)
            self.syncbn_collectives_per_step = 0
            for g in self._groups.values():
                g.todo, g.finish = 0, []
            self.model.__dict__["_hip_bn_epoch"] = self.model.__dict__.get("_hip_bn_epoch", 0) + 1
        else:
            # eval-mode scale/shift are cached until the weights, the running statistics (torch-side
            # version counters) or a HIP training step (epoch counter on the model) change them
            sig2 = (sig, self.model.__dict__.get("_hip_bn_epoch", 0),
                    tuple(b._version for b in self.model.buffers()))
            if sig2 != getattr(self, "_eval_sig", None):
                self._eval_sig = sig2
                self._eval_epoch += 1
        return x.contiguous()

    def _features(self, x):
        m = self.model
        if self.kind == "psp":
            use = m.use_ppm
            x_tmp, a, cat = self.trunk(x, 4096 if use else 2048)
            if use:
                ppm_mark = len(self.tape)
                feat = self.ppm(a, cat)
                if self.training:
                    self.tape.insert(ppm_mark, self._ppm_pool_bwd)
            else:
                feat = a
        else:
            from .psa_engine import psa_forward
            x_tmp, a, cat = self.trunk(x, 4096 if m.use_psa else 2048)
            feat = psa_forward(self, a, cat) if m.use_psa else a
        return x_tmp, feat

    def forward_eval(self, x):
        # validation is a natural checkpoint for the label counts of the training steps before it (no pending copy: free)
        self._label_watch().poll(self.model.cls[4].weight.shape[0], wait=True)
        x = self._begin(x)
        x_tmp, feat = self._features(x)
        scores = self.head(feat, self.model.cls, "m")
        h, w = self.out_hw()
        ncls = scores.C
        if self.model.zoom_factor != 1:
            out = torch.empty((self.N, ncls, h, w), dtype=F32, device=self.device)
            ops.bilinear_nhwc_to_nchw(scores.data, scores.ld, out, self.N, scores.H, scores.W, h, w, ncls)
        else:
            out = torch.empty((self.N, ncls, h, w), dtype=F32, device=self.device)
            ops.bilinear_nhwc_to_nchw(scores.data, scores.ld, out, self.N, scores.H, scores.W, scores.H,
                                      scores.W, ncls)
        return out

    def forward_train(self, x, y, ignore_index=255):
        x = self._begin(x)
        self._x = Act(x, self.N, self.H, self.W, 3, 3, "input")
        assert y.dtype == torch.int64 and y.is_cuda
        y = y.contiguous()
        h, w = self.out_hw()
        assert tuple(y.shape) == (self.N, h, w), "target must be [N,%d,%d]" % (h, w)
        self._label_watch().poll(self.model.cls[4].weight.shape[0])
        if not self._labels_checked or os.environ.get("SEMSEG_CHECK_LABELS") == "1":
            # torch's CrossEntropyLoss raises on class ids outside [0, C); the fused head would silently ignore
            # them.  Checked on this engine's first step (one sync), every step with SEMSEG_CHECK_LABELS=1.
            nbad = ops.label_check(y, self.model.cls[4].weight.shape[0], ignore_index)
            if nbad:
                raise IndexError("Target out of bounds: %d label(s) are neither ignore_index=%d nor in [0, %d)"
                                 % (nbad, ignore_index, self.model.cls[4].weight.shape[0]))
            self._labels_checked = True
        x_tmp, feat = self._features(x)
        scores, aux = self.heads_train(feat, x_tmp)
        main_loss, pred, self._rec_main = self.ce(scores, y, h, w, ignore_index, True, "m")
        self._label_watch().watch(self._rec_main["acc"])
        aux_loss, _, self._rec_aux = self.ce(aux, y, h, w, ignore_index, False, "a")
        return pred, main_loss, aux_loss

    def backward(self, gmain, gaux):
        """Replays the tape; on return every parameter gradient is in self.grad_views."""
        self._reset_grad_flags()
        self._f64_zero_sums()
        if self.hipri_main:
            if self._hi is None:
                self._hi = _shared(self.device, "hipri", lambda: torch.cuda.Stream(device=self.device, priority=-1))
            cur = torch.cuda.current_stream()
            ops.stream_wait(self._hi, cur)
            with torch.cuda.stream(self._hi):
                self._backward_chain(gmain, gaux)
            ops.stream_wait(cur, self._hi)
        else:
            self._backward_chain(gmain, gaux)

    def _backward_chain(self, gmain, gaux):
        self._main = torch.cuda.current_stream()
        self._run(TapeOp("ce", lambda: self.ce_bwd(self._rec_main, gmain), dict(rec=self._rec_main, gloss=gmain, gmul=1.0)))
        self._run(TapeOp("ce", lambda: self.ce_bwd(self._rec_aux, gaux), dict(rec=self._rec_aux, gloss=gaux, gmul=1.0)))
        for op in reversed(self.tape):
            self._run(op)
        if self._side_used:
            for st in self._sides:
                ops.stream_wait(torch.cuda.current_stream(), st)       # join the weight-gradient stream(s)
            self._side_used = False

    def _reset_grad_flags(self):
        pass  # Act objects are rebuilt every forward, so ginit starts False

    def _f64_zero_sums(self):
        # stats and sums share the arena; stats are dead after forward
        ops.zero_(self._f64_arena)
