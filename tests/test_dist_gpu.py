"""N>1 path on the GPU: 2 data-parallel ranks (SyncBN + bucketed gradient all-reduce + 1/world
scaling) must reproduce the single-process global-batch step — the equivalence the reference relies
on (8-rank SyncBN+DDP == single-process bs16 when labels have no ignore pixels) — and the
unchanged tool/train.py wrapping sequence (SyncBatchNorm conversion + DistributedDataParallel) must
drive the HIP model."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_eight_ranks_equal_single_process(report):
    """8 ranks x 1 image with SyncBN + gradient all-reduce == one process with 8 images
    (all eight ranks share the test box's single GPU through gloo; the code path is the one RCCL takes)."""
    tmp = tempfile.mkdtemp(prefix="semseg_dist8_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GLOBAL_BATCH="8")
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    subprocess.check_call([sys.executable, worker, tmp], env=env, timeout=600)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, tmp],
                          env=env, timeout=900)
    one = np.load(os.path.join(tmp, "rank0_of1.npz"))
    ranks = [np.load(os.path.join(tmp, "rank%d_of8.npz" % r)) for r in range(8)]
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["w"], r["w"]) and np.array_equal(ranks[0]["rv"], r["rv"])
    l8 = sum(r["losses"] for r in ranks) / 8.0
    e_step = np.abs(l8 - one["losses"]) / np.abs(one["losses"])
    e_w1 = np.abs(ranks[0]["w1"] - one["w1"]).max() / np.abs(one["w1"]).max()
    e_w = np.abs(ranks[0]["w"] - one["w"]).max() / np.abs(one["w"]).max()
    e_rv = np.abs(ranks[0]["rv"] - one["rv"]).max() / np.abs(one["rv"]).max()
    e_rm = np.abs(ranks[0]["rm"] - one["rm"]).max() / np.abs(one["rm"]).max()
    report("8-rank DP vs single process: losses per step (main, aux) %s; weights after 1 step %.2e, after 2 steps %.2e; "
           "running_var %.2e running_mean %.2e" % (np.array2string(e_step, precision=2), e_w1, e_w, e_rv, e_rm))
    # step 1 is the equivalence proper: the forward (SyncBN over 8 x 1 image) and the update (gradient all-reduce / 8)
    # agree to fp32 summation noise.  The second step starts from weights that differ by that noise, and this 57 x 57
    # toy net amplifies it (ReLU-mask flips) - measured 3.8e-4 in the loss, 3.9e-3 in the weights: loose bounds only.
    assert e_step[0].max() < 1e-6 and e_w1 < 2e-4
    assert e_step[1].max() < 5e-3 and e_w < 2e-2 and e_rv < 1e-4 and e_rm < 1e-2


def test_two_ranks_equal_single_process(report):
    tmp = tempfile.mkdtemp(prefix="semseg_dist_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    subprocess.check_call([sys.executable, worker, tmp], env=env, timeout=600)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, tmp],
                          env=env, timeout=900)
    one = np.load(os.path.join(tmp, "rank0_of1.npz"))
    r0 = np.load(os.path.join(tmp, "rank0_of2.npz"))
    r1 = np.load(os.path.join(tmp, "rank1_of2.npz"))
    # replicas stay identical
    assert np.array_equal(r0["w"], r1["w"]) and np.array_equal(r0["rv"], r1["rv"])
    # per-rank mean losses average to the global-batch loss (equal pixel counts)
    l2 = 0.5 * (r0["losses"] + r1["losses"])
    e_loss = np.abs(l2 - one["losses"]).max() / np.abs(one["losses"]).max()
    e_w = np.abs(r0["w"] - one["w"]).max() / np.abs(one["w"]).max()
    e_rv = np.abs(r0["rv"] - one["rv"]).max() / np.abs(one["rv"]).max()
    e_rm = np.abs(r0["rm"] - one["rm"]).max() / np.abs(one["rm"]).max()
    report("2-rank DP vs single process: loss %.2e weights-after-2-steps %.2e running_var %.2e running_mean %.2e"
           % (e_loss, e_w, e_rv, e_rm))
    # weights: two fp32 runs with different batch splits differ by ReLU-mask flips (see test_model_gpu.run_case)
    assert e_loss < 1e-5 and e_w < 2e-3 and e_rv < 1e-4 and e_rm < 1e-4
    # SyncBN exchanges per step: PSPNet-50 has 61 BatchNorm layers; bn3 + downsample BN of the 4 projection blocks, the
    # 4 PPM branches and the 2 heads share an all-reduce each -> 61 - 4 - 3 - 1 = 53 per pass, forward + backward
    assert int(r0["ncoll"]) == 106 and int(one["ncoll"]) == 0, (r0["ncoll"], one["ncoll"])


def test_reference_train_wrapping_sequence(report):
    """tool/train.py:124-157,269-276 verbatim on one rank: param groups -> SGD ->
    convert_sync_batchnorm -> DistributedDataParallel -> forward/backward/step."""
    import torch.distributed as dist
    from torch import nn
    from model.pspnet import PSPNet
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), world_size=1, rank=0)
    try:
        torch.cuda.set_device(0)
        criterion = nn.CrossEntropyLoss(ignore_index=255)
        model = PSPNet(layers=50, classes=21, zoom_factor=8, criterion=criterion, pretrained=False)
        modules_ori = [model.layer0, model.layer1, model.layer2, model.layer3, model.layer4]
        modules_new = [model.ppm, model.cls, model.aux]
        params_list = [dict(params=m.parameters(), lr=0.01) for m in modules_ori]
        params_list += [dict(params=m.parameters(), lr=0.1) for m in modules_new]
        optimizer = torch.optim.SGD(params_list, lr=0.01, momentum=0.9, weight_decay=1e-4)
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)
        model = torch.nn.parallel.DistributedDataParallel(model.cuda(), device_ids=[0])
        g = torch.Generator().manual_seed(5)
        x = torch.randn(2, 3, 73, 73, generator=g).cuda(non_blocking=True)
        y = torch.randint(0, 21, (2, 73, 73), generator=g).cuda(non_blocking=True)
        model.train()
        losses = []
        for _ in range(3):
            output, main_loss, aux_loss = model(x, y)
            loss = main_loss + 0.4 * aux_loss
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            losses.append(loss.item())
        assert output.shape == (2, 73, 73) and output.dtype == torch.int64
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
        # checkpoint round trip with the `module.` prefix (tool/train.py:234, tool/test.py:108-113)
        sd = model.state_dict()
        assert all(k.startswith("module.") for k in sd)
        m2 = torch.nn.DataParallel(PSPNet(layers=50, classes=21, zoom_factor=8, pretrained=False)).cuda()
        m2.load_state_dict(sd, strict=False)
        m2.eval()
        model.eval()
        with torch.no_grad():
            a, b = model(x), m2(x)
        assert a.shape == (2, 21, 73, 73) and torch.equal(a, b)
        report("reference wrapping sequence (SyncBN convert + DDP + SGD groups): losses %s" %
               ["%.4f" % v for v in losses])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lagging", ["side", "main"])
def test_bucket_collectives_wait_for_both_gradient_streams(lagging, report):
    """Gradients of one all-reduce bucket are written by two HIP streams (main: BN parameter gradients, stem
    and large-grid weight gradients; side: weight gradients of small grids).  A collective is ordered after the
    stream that is current when it is issued, so the trainer joins the other stream first
    (Engine.order_after_all_producers).  tests/bucket_order_worker.py replaces the collective by a snapshot
    taken with that ordering rule and slows one stream down with device sleeps.
    * with the join: no snapshot may be stale, whichever stream lags;
    * negative control (join disabled = the code before the fix): buckets completed by a side-stream weight
      gradient are still clean (every side launch already waits on the main stream), but the LAST bucket is
      completed by the stem weight gradient on the main stream, and with the side stream lagging its snapshot
      is stale (786 432 of 794 304 elements) - that was a real bug of the N > 1 path.
    Runs in a fresh process: in a long-lived one, streams can alias onto one hardware queue and hide the race."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bucket_order_worker.py"), lagging],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    report("bucket ordering, %s stream lagging: issued from %s; stale elements per bucket with the join %s, "
           "without it (control) %s" % (lagging, d["issued_from"], d["stale_with_join"], d["stale_without_join"]))
    assert sum(d["stale_with_join"]) == 0
    assert d["issued_from"][-1] == "main" and "side" in d["issued_from"]
    assert sum(d["stale_without_join"][:-1]) == 0
    if lagging == "side":
        assert d["stale_without_join"][-1] > 0, "control did not expose the race"
    else:
        assert d["stale_without_join"][-1] == 0


@pytest.mark.parametrize("xchg", ["0", "1"])
def test_two_ranks_replayed_step_plan(xchg, report):
    """The N > 1 step under the step plan (semseg_amd/plan.py): C segments between the collectives, the SyncBN exchanges and
    the gradient-bucket all-reduces re-issued as host operations in the recorded order.  Six steps on two ranks (2 eager, 2
    recorded, 2 replayed) against the same six steps sequenced launch by launch (SEMSEG_STEP_PLAN=0): the replicas of the
    replayed run stay BIT-identical (a dropped or reordered collective would split them).  The criterion for "the replay is
    the step" is structural, not numerical: the Trainer accepts a record only when the NEXT step, issued launch by launch,
    records the same calls with the same arguments and the host operations at the same places (semseg_plan_compare; the log
    must say "verified").  A numerical bound cannot be sharp here: this 57 x 57 toy net at batch 1 per rank amplifies the
    run-to-run atomics noise so much that steps 1 and 2 — the SAME launch-by-launch code in both runs — already differ by
    1e-6 and 1e-4 between them at lr 1e-4 (5e-7 and 2.4e-3 at lr 1e-2), so the losses only get a sanity bound."""
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    res = {}
    for mode in ("0", "1"):
        tmp = tempfile.mkdtemp(prefix="semseg_plan_dist%s_" % mode)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", STEPS="6", LR="1e-4", SEMSEG_STEP_PLAN=mode, SEMSEG_SYNCBN_XCHG=xchg)
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, tmp],
                              env=env, timeout=900)
        res[mode] = [np.load(os.path.join(tmp, "rank%d_of2.npz" % k)) for k in range(2)]
    p0, p1 = res["1"]
    log = str(p0["plan_log"])
    assert "recorded:" in log and "host operations" in log, log
    nhost = int(log.split("segments, ")[1].split(" host")[0])
    if xchg == "0":
        assert nhost >= int(p0["ncoll"]) + 2, log       # every SyncBN exchange (c10d) + the gradient buckets + the wait before SGD
    else:
        # the peer-memory exchange is a recorded launch (its exchange number lives in device memory): only the gradient buckets
        # and the wait before SGD are host operations
        assert 2 <= nhost <= 16, log
    assert np.array_equal(p0["w"], p1["w"]) and np.array_equal(p0["rv"], p1["rv"]) and np.array_equal(p0["rm"], p1["rm"])
    e0 = res["0"][0]
    e_first = np.abs(p0["losses"][:1] - e0["losses"][:1]).max()
    e_loss = np.abs(p0["losses"] - e0["losses"]).max(axis=1) / np.abs(e0["losses"]).max()
    report("2-rank step plan, SyncBN through %s (%s) vs launch-by-launch, lr 1e-4: replicas bit-identical after 2 replayed steps; "
           "losses per step %s" % ("c10d" if xchg == "0" else "the peer-memory exchange", log, " ".join("%.1e" % v for v in e_loss)))
    assert "verified" in log
    assert e_first == 0.0 and e_loss.max() < 5e-2


def test_syncbn_peer_memory_exchange(report):
    """The opt-in SyncBN exchange through IPC-mapped fine-grained memory (csrc/xchg.hip, SEMSEG_SYNCBN_XCHG=1)
    with 2 and 4 processes on the test box's ONE GPU: (1) the raw exchange == rank-ordered fp64 sum of the ranks' vectors, bit
    for bit, on every rank, for the vector sizes / slot counts the engine uses, plus 200 back-to-back exchanges with no host
    synchronisation; (2) a 2-rank training run with the exchange in place of the c10d all-reduce: first-step losses bit-identical
    to the c10d run (a + b is the same in either order), replicas bit-identical, later quantities within the run-to-run noise
    of the two backward kernels that merge with fp32 atomics (bounds of test_eight_ranks_equal_single_process).  If the kernels of two processes
    cannot be co-resident on this box the exchange gives up after ~1 s (bounded spins) and the test is skipped with that
    reason — the path is opt-in either way."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "xchg_worker.py")
    for W in (2, 4):
        tmp = tempfile.mkdtemp(prefix="semseg_xchg%d_" % W)
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % W,
                            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, tmp],
                           env=env, timeout=600, capture_output=True, text=True)
        if "TIMEOUT in exchange" in p.stdout:
            pytest.skip("peer-memory exchange: kernels of %d processes were not co-resident on this GPU (bounded spin gave up)" % W)
        assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
        r = [np.load(os.path.join(tmp, "xchg_rank%d_of%d.npz" % (k, W))) for k in range(W)]
        report("peer-memory SyncBN exchange, %d processes on one GPU: %d exchanges per rank, all bit-identical to the "
               "rank-ordered fp64 sum" % (W, int(r[0]["nex"])))
    worker = os.path.join(ROOT, "tests", "dist_worker.py")
    res = {}
    for mode in ("0", "1"):
        tmp = tempfile.mkdtemp(prefix="semseg_xchg_train%s_" % mode)
        subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, tmp],
                              env=dict(env, SEMSEG_SYNCBN_XCHG=mode), timeout=900)
        res[mode] = [np.load(os.path.join(tmp, "rank%d_of2.npz" % k)) for k in range(2)]
    x0, x1 = res["1"]
    # the replicas of the exchange run stay bit-identical (every rank sums the slots in rank order)
    assert np.array_equal(x0["w"], x1["w"]) and np.array_equal(x0["rv"], x1["rv"]) and np.array_equal(x0["rm"], x1["rm"])
    assert int(x0["ncoll"]) == int(res["0"][0]["ncoll"])
    e = {}
    for k in range(2):
        c, x = res["0"][k], res["1"][k]
        # the first step's losses depend on the forward only (initial weights + SyncBN statistics): a + b is the same sum in
        # either order, so they are bit-identical to the c10d run; everything after the first update also carries the fp32
        # atomics of two backward kernels (bilinear_bwd of the PPM branches, stem_wgrad), which differ run to run
        assert np.array_equal(c["losses"][0], x["losses"][0]), (k, c["losses"][0], x["losses"][0])
        e[k] = (np.abs(c["w1"] - x["w1"]).max() / np.abs(c["w1"]).max(), np.abs(c["w"] - x["w"]).max() / np.abs(c["w"]).max(),
                np.abs(c["losses"][1] - x["losses"][1]).max() / np.abs(c["losses"][1]).max(),
                np.abs(c["rv"] - x["rv"]).max() / np.abs(c["rv"]).max())
        assert e[k][0] < 2e-4 and e[k][1] < 2e-2 and e[k][2] < 5e-3 and e[k][3] < 1e-4, e[k]
    report("2-rank training, SyncBN statistics through the peer-memory exchange instead of c10d (%d exchanges per step): "
           "first-step losses bit-identical, replicas bit-identical; vs the c10d run weights after 1 / 2 steps %.1e / %.1e, "
           "second-step losses %.1e, running_var %.1e (run-to-run fp32-atomics noise)"
           % (int(x0["ncoll"]), e[0][0], e[0][1], e[0][2], e[0][3]))
