"""In-situ backward parity (test infrastructure).

`InsituChecker` is installed as `Engine.tape_hook`.  For every backward step of a real train step it
snapshots the HIP path's OWN operands (x, dy, the saved mean / invstd, the post-activation tensor that carries the
ReLU mask, labels, ...) right before the launch, lets the HIP kernels run, and then recomputes that single op's
backward on the CPU twice from those same operands: in fp64 (the reference value) and in fp32 (torch's CPU kernels
= the arithmetic the reference itself runs, its noise level).  Because every op is re-derived from the operands the
HIP path actually used, ReLU-mask flips and the conditioning of the network do not enter: a systematic error of
1e-5 in any single dgrad / wgrad / BN-backward / CE-backward kernel shows up as a ratio >> 1.

Criterion per quantity, both against the fp64 recomputation:
    rms error   ||a - ref||_2 / ||ref||_2   :  hip <= RATIO     * cpu_fp32 + FLOOR      (the noise LEVEL)
    max error   max|a - ref| / max|ref|     :  hip <= RATIO_MAX * cpu_fp32 + FLOOR      (no outlier element)
The rms is the statistic with the 3x bound: the maximum of ~1e6 rounding errors is itself a noisy sample (two fp32
implementations of one sum differ by up to ~2.6x in it on the 73x73 case), so it gets the looser outlier bound.

Reference formulas (file:line under /root/reference): conv backward = adjoints of nn.Conv2d (model/resnet.py:63-69,
model/pspnet.py:65-77); BatchNorm backward = torch's batch_norm_backward for model/resnet.py:76-92 (train mode, biased
variance); bilinear align_corners=True adjoint (model/pspnet.py:25,95,100); AdaptiveAvgPool2d adjoint
(model/pspnet.py:14); MaxPool2d(3,2,1) adjoint (model/resnet.py:115); CrossEntropyLoss(ignore_index) of the upsampled
scores (model/pspnet.py:95-102, tool/train.py:121); PSA contraction / softmax / psamask adjoints
(model/psanet.py:80-91, lib/psa/src/cpu/psamask.cpp:59-113).
"""
import os

import torch
import torch.nn.functional as F

RATIO = 3.0
RATIO_MAX = 5.0
# Convs that run the Winograd F(2x2, 3x3) path (round 3): every output is recombined from 9 of the 16 element-wise
# products with alternating signs, each about as large as the result, so the rounding noise of a sum of like-sized terms
# is amplified by ~sqrt(9) = 3 over the direct convolution's (measured when the path was first run through this test:
# rms 1.5 - 4.1 x the CPU-fp32 error on dgrad, 1.8 x on wgrad; relative rms error 7e-7).  Their bound is the direct
# kernels' bound (which sits at 2.0 - 2.5 x for the worst direct layer) times that factor, rounded down: 6 x rms,
# 10 x max-abs.  Quantities are reported as "dgrad-wino" / "wgrad-wino" so the two populations stay separate.
RATIO_WINO = 6.0
RATIO_MAX_WINO = 10.0
FLOOR = 2e-7   # two fp32 ulps of the largest element: ops that are exact on the CPU (pure routing) have err_cpu = 0


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup v2 cpu.max quota) — the GPU box shows 256
    hardware threads in the affinity mask but grants 16; oversubscribing OpenMP by 16x stalls every CPU reference."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, torch.get_num_threads() if n > 64 else n))


def _nchw(t, C):
    """NHWC device view (any channel stride) -> contiguous NCHW CPU fp32."""
    return t[..., :C].detach().permute(0, 3, 1, 2).contiguous().cpu()


def conv_bwd_taps(x, w_shape, w, dy, stride, pad, dil, want_dx=True):
    """fp64 conv backward as one GEMM per tap (multi-threaded BLAS): torch's own fp64 conv backward on the CPU
    parallelises over the batch only, which is 2 here.  x [N,Ci,H,W], dy [N,Co,Ho,Wo] (any float dtype) ->
    (dW [Co,Ci,R,S], dX or None) in fp64.  Checked against torch.nn.grad in tests/test_insitu_refs_cpu.py."""
    Co, Ci, R, S = w_shape
    N, _, H, W = x.shape
    Ho, Wo = dy.shape[2], dy.shape[3]
    xp = F.pad(x.double(), (pad, pad, pad, pad))
    dyf = dy.double().permute(1, 0, 2, 3).reshape(Co, N * Ho * Wo)                # [Co, NL]
    dW = torch.empty(Co, Ci, R, S, dtype=torch.float64)
    dxp = torch.zeros_like(xp) if want_dx else None
    wd = None if w is None else w.double()
    for r in range(R):
        for s_ in range(S):
            h0, w0 = r * dil, s_ * dil
            sl = (slice(None), slice(None), slice(h0, h0 + (Ho - 1) * stride + 1, stride),
                  slice(w0, w0 + (Wo - 1) * stride + 1, stride))
            xs = xp[sl].permute(1, 0, 2, 3).reshape(Ci, N * Ho * Wo)              # [Ci, NL]
            dW[:, :, r, s_] = dyf @ xs.t()
            if want_dx:
                g = (wd[:, :, r, s_].t() @ dyf).reshape(Ci, N, Ho, Wo).permute(1, 0, 2, 3)
                dxp[sl] += g
    dx = dxp[:, :, pad:pad + H, pad:pad + W].contiguous() if want_dx else None
    return dW, dx


def _err(a, ref):
    """(max-abs error / max|ref|, rms error / rms ref)"""
    ref = ref.double()
    diff = a.double() - ref
    d = float(ref.abs().max())
    r = float(ref.pow(2).mean().sqrt())
    return float(diff.abs().max()) / max(d, 1e-300), float(diff.pow(2).mean().sqrt()) / max(r, 1e-300)


class InsituChecker:
    def __init__(self, eng, log=print, only=None):
        self.eng = eng
        self.log = log
        self.only = only     # callable(kind, module name or None) -> bool: ops it rejects run unchecked (batch-16 sampling)
        self.rows = []       # (kind, name, quantity, max_hip, max_cpu32, rms_hip, rms_cpu32)
        self.names = {m: n for n, m in eng.model.named_modules()}
        torch.set_num_threads(usable_cores())

    # ------------------------------------------------------------------ hook
    def __call__(self, op):
        h = getattr(self, "_chk_" + op.kind, None)
        if h is not None and self.only is not None:
            mod = op.ctx.get("m") or op.ctx.get("bm")
            if not self.only(op.kind, self.names.get(mod) if mod is not None else None):
                h = None
        if h is None:
            op.fn()
            return
        torch.cuda.synchronize()
        h(op)

    def _launch(self, op):
        op.fn()
        torch.cuda.synchronize()

    def _rec(self, kind, name, qty, hip, ref64, c32):
        mh, rh = _err(hip, ref64)
        mc, rc = _err(c32, ref64)
        self.rows.append((kind, name, qty, mh, mc, rh, rc))

    def failures(self):
        def ok(r):
            wino = r[2].endswith("-wino")
            kr, km = (RATIO_WINO, RATIO_MAX_WINO) if wino else (RATIO, RATIO_MAX)
            return r[5] <= kr * r[6] + FLOOR and r[3] <= km * r[4] + FLOOR
        return [r for r in self.rows if not ok(r)]

    def summary(self):
        by = {}
        for kind, name, qty, mh, mc, rh, rc in self.rows:
            d = by.setdefault((kind, qty), [0, 0.0, 0.0, 0.0, "", 0.0])
            d[0] += 1
            d[1] = max(d[1], rh)
            d[2] = max(d[2], rc)
            r = rh / max(rc, FLOOR / RATIO)
            if r > d[3]:
                d[3], d[4] = r, name
            d[5] = max(d[5], mh / max(mc, FLOOR / RATIO))
        out = []
        for (kind, qty), (n, eh, ec, r, name, rm) in sorted(by.items()):
            out.append("  %-10s %-8s n=%3d  rms err hip %.2e cpu-fp32 %.2e  worst rms ratio %.2f (%s), worst max-abs ratio %.2f"
                       % (kind, qty, n, eh, ec, r, name, rm))
        return "\n".join(out)

    # ------------------------------------------------------------------ conv: dgrad, wgrad, bias grad
    def _chk_conv(self, op):
        x, y, cl, m = op.ctx["x"], op.ctx["y"], op.ctx["cl"], op.ctx["m"]
        name = self.names[m]
        xs = _nchw(x.data, x.C)
        dy = _nchw(y.grad, y.C)
        w = m.weight.detach().cpu()
        has_dx = x.name != "input"
        gx0 = _nchw(x.grad, x.C) if (has_dx and x.ginit) else None
        self._launch(op)
        kw = dict(stride=cl.stride, padding=cl.pad, dilation=cl.dil)
        gw64, gx64 = conv_bwd_taps(xs, w.shape, w, dy, cl.stride, cl.pad, cl.dil, want_dx=has_dx)
        gw32 = torch.nn.grad.conv2d_weight(xs, w.shape, dy, **kw)
        wino = "-wino" if getattr(cl, "wino", None) is not None else ""
        self._rec("conv", name, "wgrad" + wino, cl.wgrad.detach().cpu(), gw64, gw32)
        if m.bias is not None:
            self._rec("conv", name, "bgrad", cl.bgrad.detach().cpu(), dy.double().sum((0, 2, 3)), dy.sum((0, 2, 3)))
        if has_dx:
            gx32 = torch.nn.grad.conv2d_input(xs.shape, w, dy, **kw)
            if gx0 is not None:
                gx64 = gx64 + gx0.double()
                gx32 = gx32 + gx0
            qty = "dgrad" + wino
            if x.bn_reduced:
                # this launch also did the BatchNorm-backward reduction of the layer that produced x: what it stored is
                # g = dx * (x > 0); the sums it accumulated are checked by the BatchNorm entry (dgamma / dbeta / dy)
                qty = "dgrad+bnr" + wino
                if x.bnsrc["relu"]:
                    gx64 = gx64 * (xs > 0)
                    gx32 = gx32 * (xs > 0)
            self._rec("conv", name, qty, _nchw(x.grad, x.C), gx64, gx32)

    # ------------------------------------------------------------------ BatchNorm (+ReLU, residual, downsample BN)
    @staticmethod
    def _bn_bwd(g, yy, mean, invstd, gamma, cnt, dt):
        g, yy = g.to(dt), yy.to(dt)
        mean, invstd, gamma = (t.to(dt).view(1, -1, 1, 1) for t in (mean, invstd, gamma))
        xh = (yy - mean) * invstd
        db = g.sum((0, 2, 3))
        dg = (g * xh).sum((0, 2, 3))
        dy = gamma * invstd * (g - db.view(1, -1, 1, 1) / cnt - xh * dg.view(1, -1, 1, 1) / cnt)
        return dy, dg, db

    def _chk_bn_act(self, op):
        c = op.ctx
        y, bm, bl, out, cnt = c["y"], c["bm"], c["bl"], c["out"], c["cnt"]
        name = self.names[bm]
        assert self.eng.world == 1 or not self.eng.dist_on, "in-situ check is single-process"
        C = y.C
        dout = _nchw(out.grad, C)
        o = _nchw(out.data, C)
        yy = _nchw(y.data, C)
        dm = None if c["dropmask"] is None else c["dropmask"].detach().cpu()
        mean, invstd = bl.mean.detach().cpu(), bl.invstd.detach().cpu()
        gamma = bm.weight.detach().cpu()
        y2 = c["y2"]
        if y2 is not None:
            yy2 = _nchw(y2.data, C)
            bl2, bm2 = c["bl2"], c["bm2"]
            mean2, invstd2, gamma2 = bl2.mean.detach().cpu(), bl2.invstd.detach().cpu(), bm2.weight.detach().cpu()
        self._launch(op)

        def masked(dt):
            g = dout.to(dt)
            if dm is not None:
                g = g * dm.to(dt)[:, :, None, None]
            if c["relu"]:
                g = g * (o > 0).to(dt)
            return g
        g64, g32 = masked(torch.float64), masked(torch.float32)
        r64 = self._bn_bwd(g64, yy, mean, invstd, gamma, cnt, torch.float64)
        r32 = self._bn_bwd(g32, yy, mean, invstd, gamma, cnt, torch.float32)
        self._rec("bn_act", name, "dy", _nchw(y.grad, C), r64[0], r32[0])
        self._rec("bn_act", name, "dgamma", bl.ggrad.detach().cpu(), r64[1], r32[1])
        self._rec("bn_act", name, "dbeta", bl.bgrad.detach().cpu(), r64[2], r32[2])
        if c["res"] is not None:
            self._rec("bn_act", name, "dres", _nchw(c["res"].grad, C), g64, g32)
        if y2 is not None:
            q64 = self._bn_bwd(g64, yy2, mean2, invstd2, gamma2, cnt, torch.float64)
            q32 = self._bn_bwd(g32, yy2, mean2, invstd2, gamma2, cnt, torch.float32)
            n2 = self.names[bm2]
            self._rec("bn_act", n2, "dy", _nchw(y2.grad, C), q64[0], q32[0])
            self._rec("bn_act", n2, "dgamma", bl2.ggrad.detach().cpu(), q64[1], q32[1])
            self._rec("bn_act", n2, "dbeta", bl2.bgrad.detach().cpu(), q64[2], q32[2])

    # ------------------------------------------------------------------ stem conv (NCHW input, 3 -> 64, stride 2)
    def _chk_stem_wgrad(self, op):
        x, y, m = op.ctx["x"], op.ctx["y"], op.ctx["m"]
        xs = x.detach().cpu()
        dy = _nchw(y.grad, y.C)
        self._launch(op)
        kw = dict(stride=2, padding=1, dilation=1)
        g64, _ = conv_bwd_taps(xs, m.weight.shape, None, dy, 2, 1, 1, want_dx=False)
        g32 = torch.nn.grad.conv2d_weight(xs, m.weight.shape, dy, **kw)
        self._rec("stem", self.names[m], "wgrad", self.eng.grad_views[m.weight].detach().cpu(), g64, g32)

    # ------------------------------------------------------------------ max pool 3x3 / 2 / 1
    def _chk_maxpool(self, op):
        x, y = op.ctx["x"], op.ctx["y"]
        xs = _nchw(x.data, x.C)
        dy = _nchw(y.grad, y.C)
        self._launch(op)
        dx = _nchw(x.grad, x.C)

        def ref(dt):
            xx = xs.to(dt).requires_grad_(True)
            return torch.autograd.grad(F.max_pool2d(xx, 3, 2, 1), xx, dy.to(dt))[0]
        r64, r32 = ref(torch.float64), ref(torch.float32)
        # windows whose maximum is attained more than once (post-ReLU zeros) may route to either position:
        # compare outside of them, and check conservation of the total inside
        mx = F.max_pool2d(xs, 3, 2, 1)
        pad = F.pad(xs, (1, 1, 1, 1), value=float("-inf"))
        win = pad.unfold(2, 3, 2).unfold(3, 3, 2)                         # [N,C,Ho,Wo,3,3]
        ties = ((win == mx[..., None, None]).sum((-1, -2)) > 1).float()    # [N,C,Ho,Wo]
        Cc = xs.shape[1]
        spread = F.conv_transpose2d(ties, torch.ones(Cc, 1, 3, 3), stride=2, padding=1, groups=Cc)
        assert spread.shape == xs.shape
        keep = (spread == 0)
        self._rec("maxpool", "layer0.maxpool", "dx", dx * keep, r64 * keep, r32 * keep)
        self._rec("maxpool", "layer0.maxpool", "sum", dx.double().sum((2, 3)), r64.sum((2, 3)), r32.sum((2, 3)))

    # ------------------------------------------------------------------ bilinear (align_corners=True) adjoint
    def _chk_upsample(self, op):
        x, Ho, Wo = op.ctx["x"], op.ctx["Ho"], op.ctx["Wo"]
        C = x.C
        dy = _nchw(op.ctx["dy"](), C)
        self._launch(op)

        def ref(dt):
            xx = torch.zeros(x.N, C, x.H, x.W, dtype=dt, requires_grad=True)
            up = F.interpolate(xx, (Ho, Wo), mode="bilinear", align_corners=True)
            return torch.autograd.grad(up, xx, dy.to(dt))[0]
        self._rec("upsample", "%s %dx%d->%dx%d" % (x.name, x.H, x.W, Ho, Wo), "dx", _nchw(x.grad, C),
                  ref(torch.float64), ref(torch.float32))

    # ------------------------------------------------------------------ PPM adaptive average pools (all bins at once)
    def _chk_ppm_pool(self, op):
        cat, dpool, bins, C = op.ctx["cat"], op.ctx["dpool"], op.ctx["bins"], op.ctx["C"]
        N, H, W = cat.N, cat.H, cat.W
        g0 = _nchw(cat.grad, C)
        dp = dpool.detach().cpu()
        self._launch(op)
        g1 = _nchw(cat.grad, C)

        def ref(dt):
            acc = g0.to(dt).clone()
            off = 0
            for b in bins:
                n = N * b * b * C
                d = dp[off:off + n].view(N, b, b, C).permute(0, 3, 1, 2).to(dt)
                off += n
                xx = torch.zeros(N, C, H, W, dtype=dt, requires_grad=True)
                acc = acc + torch.autograd.grad(F.adaptive_avg_pool2d(xx, b), xx, d)[0]
            return acc
        self._rec("ppm_pool", "ppm", "dx", g1, ref(torch.float64), ref(torch.float32))

    # ------------------------------------------------------------------ fused upsample + CE head
    def _chk_ce(self, op):
        rec, gloss = op.ctx["rec"], op.ctx["gloss"]
        s = rec["scores"]
        sc = _nchw(s.data, s.C)
        lab = rec["label"].detach().cpu()
        gl = float(gloss.detach().cpu().reshape(-1)[0])
        self._launch(op)

        def ref(dt):
            z = sc.to(dt).requires_grad_(True)
            up = F.interpolate(z, (rec["H"], rec["W"]), mode="bilinear", align_corners=True)
            loss = F.cross_entropy(up, lab, ignore_index=rec["ignore"])
            return torch.autograd.grad(loss, z)[0] * gl
        self._rec("ce", "scores[%d classes]" % s.C, "dscores", _nchw(s.grad, s.C), ref(torch.float64),
                  ref(torch.float32))

    # ------------------------------------------------------------------ PSA contraction + softmax + psamask adjoints
    def _chk_psa_contract(self, op):
        from oracle import segnet
        from oracle import psamask as pm
        c = op.ctx
        xs, ym, aff, zcat, zoff, typ, psa, P = c["xs"], c["ym"], c["aff"], c["zcat"], c["zoff"], c["typ"], c["psa"], c["P"]
        h, w, alpha = c["h"], c["w"], c["alpha"]
        N, C, hw = xs.N, xs.C, h * w
        gz = zcat.grad[..., zoff:zoff + C].detach().reshape(N, hw, C).cpu()        # [n,q,c]
        xv = xs.data[..., :C].detach().reshape(N, hw, C).cpu()                    # [n,p,c]
        A = aff[:N * hw, :hw].detach().reshape(N, hw, hw).cpu()                   # [n,q,p]
        gxs0 = xs.grad[..., :C].detach().reshape(N, hw, C).cpu() if xs.ginit else None
        self._launch(op)

        def ref(dt):
            g, x_, a = gz.to(dt), xv.to(dt), A.to(dt)
            dx = torch.einsum("nqp,nqc->npc", a, g)
            if gxs0 is not None:
                dx = dx + gxs0.to(dt)
            dA = torch.einsum("nqc,npc->nqp", g, x_)
            if psa.psa_softmax:
                sm = a / alpha
                t = dA * alpha
                draw = sm * (t - (t * sm).sum(-1, keepdim=True))
            else:
                draw = dA * alpha
            if psa.compact:
                dm = draw if typ == 0 else draw.transpose(1, 2)                   # [n, pixel, tap]
                return dx, dm.reshape(N, h, w, hw)
            dref = draw.transpose(1, 2).reshape(N, hw, h, w).contiguous()         # reference layout [N, HW, H, W]
            dmask = segnet._perm(pm.psa_mask_backward, dref, typ, psa.mask_h, psa.mask_w)   # [N, taps, h, w]
            return dx, dmask.permute(0, 2, 3, 1)
        r64, r32 = ref(torch.float64), ref(torch.float32)
        nm = "psa branch %d" % typ
        self._rec("psa", nm, "dx", xs.grad[..., :C].detach().reshape(N, hw, C).cpu(), r64[0], r32[0])
        self._rec("psa", nm, "dmask", ym.grad[..., :ym.C].detach().cpu(), r64[1], r32[1])


def run_insitu(model, x, y, log=print, only=None):
    """One train step of `model` (cuda, train mode) with the checker installed; returns the checker."""
    from semseg_amd.engine import Engine
    eng = Engine(model, x.shape[0], x.shape[2], x.shape[3], True, model.kind)
    chk = InsituChecker(eng, log, only)
    pred, ml, al = eng.forward_train(x, y, 255)
    eng.tape_hook = chk
    g_main = torch.ones(1, device=x.device)
    g_aux = torch.full((1,), 0.4, device=x.device)
    eng.backward(g_main, g_aux)
    torch.cuda.synchronize()
    eng.tape_hook = None
    return chk, float(ml.item()), float(al.item())
