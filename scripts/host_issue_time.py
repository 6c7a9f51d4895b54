"""Host-side issue time of a train step against its GPU time, for the two step drivers:
  eager  Python sequences every launch through ctypes (rounds 1-4)
  plan   recorded once, replayed by one semseg_plan_replay call (csrc/plan.hip)
(a third, one hipGraph of the record, was measured in rounds 5-6 and removed: profiles/r06_host_issue_b2.json)
host ms = wall time of the python thread inside Trainer.step with an idle queue in front of it (synchronize before every
step: nothing to wait for but the issue itself); device ms = back-to-back steps.   python scripts/host_issue_time.py [batch] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model.pspnet import PSPNet
from semseg_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = {"batch": B}
# the N > 1 code path on one rank: SEMSEG_FORCE_DIST=1 python -m torch.distributed.run --nproc-per-node 1 ... host_issue_time.py
DIST = os.environ.get("SEMSEG_FORCE_DIST") == "1" and "RANK" in os.environ
if DIST:
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    res["path"] = "N > 1 code path on one rank, SyncBN exchange " + os.environ.get("SEMSEG_SYNCBN_XCHG", "0")
for mode in ("eager", "plan"):
    torch.manual_seed(0)
    m = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False).cuda().train()
    tr = Trainer(m, base_lr=0.01, sync_bn=True)
    tr.use_plan = mode != "eager"
    x = torch.randn(B, 3, 473, 473).cuda()
    y = torch.randint(0, 150, (B, 473, 473)).cuda()
    for _ in range(6):
        tr.step(x, y, 0.01)
    torch.cuda.synchronize()
    host = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.step(x, y, 0.01)
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tr.step(x, y, 0.01)
    torch.cuda.synchronize()
    dev = (time.perf_counter() - t0) / 10 * 1e3
    res[mode] = {"host_issue_ms": round(min(host), 3), "host_issue_ms_all": [round(h, 3) for h in host],
                 "step_ms_back_to_back": round(dev, 3), "plan_log": tr.plan_log[-1:]}
    print(mode, res[mode], flush=True)
    del tr, m
    torch.cuda.empty_cache()
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res))
