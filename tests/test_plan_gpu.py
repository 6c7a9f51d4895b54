"""Step plan on the GPU: a train step recorded once and replayed from C (semseg_plan_replay) — and, single-GPU, from one
hipGraph — must be the step the launch-by-launch driver runs: tool/train.py:269-276 with a different batch, learning rate
(poly schedule) and Dropout2d mask every step.  Two eager runs of the same eight steps differ by the run-to-run noise of the
kernels that merge with fp32 / fp64 atomics; the replayed runs must sit inside 4x that noise (+ 1e-6) on every loss and on the
final weights.  Also: a plan whose arenas moved is re-recorded, and the kernel-timing path bypasses the plan."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
STEPS = 8


def _run(mode, arch="psp"):
    from oracle import segnet
    from semseg_amd.trainer import Trainer, poly_learning_rate
    torch.manual_seed(5)
    if arch == "psp":
        from model.pspnet import PSPNet
        m = PSPNet(layers=50, classes=21, zoom_factor=8, dropout=0.1, pretrained=False)
        size = 73
    else:
        from model.psanet import PSANet
        m = PSANet(layers=50, classes=19, zoom_factor=8, dropout=0.1, pretrained=False, psa_type=2, compact=False,
                   shrink_factor=2, mask_h=9, mask_w=9, normalization_factor=1.0, psa_softmax=True)
        size = 65
    ncls = m.cls[4].weight.shape[0]
    m.load_state_dict(segnet.recipe_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234))
    m = m.cuda().train()
    tr = Trainer(m, base_lr=0.01, sync_bn=False)
    tr.use_plan = mode != "eager"
    tr.use_graph = mode == "graph"
    g = torch.Generator().manual_seed(11)
    losses = []
    for it in range(STEPS):
        x = torch.randn(2, 3, size, size, generator=g).cuda()           # a fresh tensor (new address) every step
        y = torch.randint(0, ncls, (2, size, size), generator=g).cuda()
        _, ml, al = tr.step(x, y, poly_learning_rate(0.01, it, 30))
        losses.append((float(ml.item()), float(al.item())))
    tr.check_labels()
    torch.cuda.synchronize()
    e = next(iter(tr.engines.values()))
    return np.array(losses), tr.flat_w.detach().cpu().numpy().copy(), tr, e


@pytest.mark.parametrize("arch", ["psp", "psa"])
def test_replayed_steps_equal_eager_steps(arch, report):
    la, wa, _, _ = _run("eager", arch)
    lb, wb, _, _ = _run("eager", arch)
    lp, wp, trp, ep = _run("plan", arch)
    lg, wg, trg, eg = _run("graph", arch)
    assert getattr(ep, "_plan", None) is not None and ep._plan_replays == STEPS - 4, trp.plan_log
    assert getattr(eg, "_plan", None) is not None and eg._plan.graph is not None and eg._plan_replays == STEPS - 4, trg.plan_log
    noise_l = np.abs(la - lb).max(axis=1) / np.abs(la).max()
    noise_w = np.abs(wa - wb).max() / np.abs(wa).max()
    rep = []
    for name, l, w in (("C replay", lp, wp), ("hipGraph", lg, wg)):
        dl = np.abs(l - la).max(axis=1) / np.abs(la).max()
        dw = np.abs(w - wa).max() / np.abs(wa).max()
        rep.append("%s: losses per step %s, final weights %.1e" % (name, " ".join("%.1e" % v for v in dl), dw))
        assert np.all(dl <= 4 * noise_l.max() + 1e-6), (name, dl, noise_l)
        assert dw <= 4 * noise_w + 1e-6, (name, dw, noise_w)
    report("step plan [%s, Dropout2d 0.1, poly lr, new input tensors every step], %d steps (2 eager, 2 recorded and compared, %d replayed): %s; "
           "%s; eager vs eager (run-to-run noise): losses %s, weights %.1e; %s"
           % (arch, STEPS, STEPS - 4, rep[0], rep[1], " ".join("%.1e" % v for v in noise_l), noise_w, trg.plan_log[-1]))


def test_plan_is_rerecorded_when_an_arena_moves_and_bypassed_by_the_kernel_timer():
    from semseg_amd import engine as E
    _, _, tr, e = _run("plan")
    assert e._plan is not None
    x = torch.randn(2, 3, 73, 73).cuda()
    y = torch.randint(0, 21, (2, 73, 73)).cuda()
    E.ARENA_GEN[0] += 1                       # what a growing Winograd arena / an evicted scratch arena does
    tr.step(x, y, 0.01)
    assert any("discarded" in l for l in tr.plan_log) and e._plan is None      # first record of the new pair
    tr.step(x, y, 0.01)
    assert e._plan is not None and e._plan_replays == 0 and "verified" in tr.plan_log[-1]
    tr.step(x, y, 0.01)
    assert e._plan_replays == 1
    kt = E.KernelTimer()
    e.ktimer = kt
    tr.step(x, y, 0.01)                       # timed launch by launch, on the caller's stream
    torch.cuda.synchronize()
    assert e._plan_replays == 1 and kt.summary()
    e.ktimer = None
    tr.step(x, y, 0.01)
    assert e._plan_replays == 2
    torch.cuda.synchronize()
