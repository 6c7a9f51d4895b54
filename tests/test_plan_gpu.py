"""Step plan on the GPU: a train step recorded once and replayed from C (semseg_plan_replay) must be the step the launch-by-launch driver runs: tool/train.py:269-276 with a different batch, learning rate
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
    assert getattr(ep, "_plan", None) is not None and ep._plan_replays == STEPS - 4, trp.plan_log
    noise_l = np.abs(la - lb).max(axis=1) / np.abs(la).max()
    noise_w = np.abs(wa - wb).max() / np.abs(wa).max()
    rep = []
    for name, l, w in (("C replay", lp, wp),):
        dl = np.abs(l - la).max(axis=1) / np.abs(la).max()
        dw = np.abs(w - wa).max() / np.abs(wa).max()
        rep.append("%s: losses per step %s, final weights %.1e" % (name, " ".join("%.1e" % v for v in dl), dw))
        assert np.all(dl <= 4 * noise_l.max() + 1e-6), (name, dl, noise_l)
        assert dw <= 4 * noise_w + 1e-6, (name, dw, noise_w)
    report("step plan [%s, Dropout2d 0.1, poly lr, new input tensors every step], %d steps (2 eager, 2 recorded and compared, %d replayed): %s; "
           "eager vs eager (run-to-run noise): losses %s, weights %.1e; %s"
           % (arch, STEPS, STEPS - 4, rep[0], " ".join("%.1e" % v for v in noise_l), noise_w, trp.plan_log[-1]))


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


def test_replay_takes_unaligned_batch_slices_and_notices_changed_host_state():
    """(1) a contiguous slice of a larger device batch is not 16-byte aligned (3 * 73 * 73 * 4 bytes per image = 12 mod 16;
    the headline 3 * 473 * 473 * 4 likewise): the replayed step must take it like the eager step does; (2) host state that a record
    bakes in (BatchNorm training flags, Dropout2d p, weight decay) changes -> the plan is discarded and recorded again."""
    _, _, tr, e = _run("plan")
    assert e._plan is not None
    big_x = torch.randn(5, 3, 73, 73).cuda()
    big_y = torch.randint(0, 21, (5, 73, 73)).cuda()
    x, y = big_x[1:3], big_y[1:3]
    assert x.is_contiguous() and x.data_ptr() % 16 != 0
    n0 = e._plan_replays
    tr.step(x, y, 0.01)
    torch.cuda.synchronize()
    assert e._plan_replays == n0 + 1
    assert torch.equal(e._plan_x, x) and torch.equal(e._plan_y, y)
    tr.wd = 5e-4
    tr.step(x, y, 0.01)
    assert e._plan is None and any("host state" in l for l in tr.plan_log)
    tr.step(x, y, 0.01)
    assert e._plan is not None and e._plan_replays == 0
    tr.model.cls[3].p = 0.3
    tr.step(x, y, 0.01)
    assert e._plan is None
    torch.cuda.synchronize()


def test_dropout_counter_of_replays_follows_the_host_counter():
    """eager steps interleaved with replays (kernel timer set) advance the host dropout counter; the next replay must
    not reuse the counter values the eager step consumed — the mask of every step is the mask of its own call number."""
    from semseg_amd import engine as E
    from semseg_amd import ops
    _, _, tr, e = _run("plan")
    x = torch.randn(2, 3, 73, 73).cuda()
    y = torch.randint(0, 21, (2, 73, 73)).cuda()

    def mask_of_call(k, n):
        m = torch.empty(n, device="cuda")
        ops.dropout2d_mask(m, 0.1, torch.initial_seed(), k, None)
        return m

    def cls_mask():
        return [t for (seq, tag), t in e._bufs.items() if tag == "dropmask"][0].reshape(-1).clone()

    tr.step(x, y, 0.01)                # replay
    torch.cuda.synchronize()
    c0 = e._drop_calls
    assert torch.equal(cls_mask(), mask_of_call(c0 - 1, cls_mask().numel()))     # cls drew call c0 - 1, aux call c0
    e.ktimer = E.KernelTimer()
    tr.step(x, y, 0.01)                # eager (timed) step in between
    e.ktimer = None
    tr.step(x, y, 0.01)                # replay again
    torch.cuda.synchronize()
    assert e._drop_calls == c0 + 4
    assert torch.equal(cls_mask(), mask_of_call(c0 + 3, cls_mask().numel()))


def test_output_lifetime_contract():
    """INTEGRATION.md "Output lifetime" (VERDICT r5 item 6): the two losses a step returns are fresh tensors — a list of them
    collected over steps keeps every step's value, as with the reference (model/pspnet.py:101-103) — on the Trainer path
    (eager, recorded and replayed steps) and on the nn.Module path; `pred` is the engine's buffer: documented as valid until the
    next step, and the test pins that it IS shared, so that the document cannot drift from the code."""
    from oracle import segnet
    from model.pspnet import PSPNet
    from semseg_amd.trainer import Trainer
    torch.manual_seed(3)
    m = PSPNet(layers=50, classes=21, zoom_factor=8, dropout=0.1, pretrained=False)
    m.load_state_dict(segnet.recipe_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=1234))
    m = m.cuda().train()
    tr = Trainer(m, base_lr=0.01, sync_bn=False)
    g = torch.Generator().manual_seed(2)
    kept, vals, preds = [], [], []
    for it in range(7):          # 2 eager + 2 recorded + 3 replayed
        x = torch.randn(2, 3, 73, 73, generator=g).cuda()
        y = torch.randint(0, 21, (2, 73, 73), generator=g).cuda()
        pred, ml, al = tr.step(x, y, 0.01)
        kept.append((ml, al))
        vals.append((float(ml.item()), float(al.item())))
        preds.append(pred)
    torch.cuda.synchronize()
    e = next(iter(tr.engines.values()))
    assert e._plan_replays == 3
    assert len({v[0] for v in vals}) == 7                                        # the steps really differ
    assert [(float(a.item()), float(b.item())) for a, b in kept] == vals         # ... and every kept loss kept its value
    assert len({p.data_ptr() for p in preds}) == 1                               # pred: the engine's buffer, as documented
    # nn.Module path
    m2 = PSPNet(layers=50, classes=21, zoom_factor=8, dropout=0.0, pretrained=False).cuda().train()
    kept, vals = [], []
    for it in range(3):
        x = torch.randn(2, 3, 73, 73, generator=g).cuda()
        y = torch.randint(0, 21, (2, 73, 73), generator=g).cuda()
        _, ml, al = m2(x, y)
        (ml + 0.4 * al).backward()
        kept.append((ml.detach(), al.detach()))
        vals.append((float(ml.item()), float(al.item())))
    assert [(float(a.item()), float(b.item())) for a, b in kept] == vals and len({v[0] for v in vals}) == 3
