"""Diagnosis of the hipGraph capture of a recorded step (csrc/plan.hip): one small model, plan recorded, capture with
SEMSEG_PLAN_DEBUG=1 in a child process per variant so that a crash of the runtime is survivable.
python scripts/graph_debug.py [child]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import faulthandler; faulthandler.enable()
    import torch
    from model.pspnet import PSPNet
    from semseg_amd.trainer import Trainer
    m = PSPNet(layers=50, classes=21, zoom_factor=8, dropout=0.1, pretrained=False).cuda().train()
    tr = Trainer(m, base_lr=0.01, sync_bn=False)
    tr.use_plan, tr.use_graph = True, True
    x = torch.randn(2, 3, 73, 73).cuda(); y = torch.randint(0, 21, (2, 73, 73)).cuda()
    for i in range(6):
        _, ml, _ = tr.step(x, y, 0.01)
        torch.cuda.synchronize()
        print("step", i, float(ml.item()), tr.plan_log[-1:] if tr.plan_log else "", flush=True)
    print("CHILD OK", flush=True)
    sys.exit(0)
for name, env in [("side stream forked from the step's stream (graph default)", {}),
                  ("high-priority chain only", {"SEMSEG_GRAPH_KEEP_HIPRI": "1", "SEMSEG_SIDE_WGRAD": "0"}),
                  ("fork of a fork: side from the high-priority chain (round-4 stream layout)", {"SEMSEG_GRAPH_KEEP_HIPRI": "1"}),
                  ("one stream", {"SEMSEG_SIDE_WGRAD": "0", "SEMSEG_HIPRI_MAIN": "0"})]:
    e = dict(os.environ, SEMSEG_PLAN_DEBUG="1", **env)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True, timeout=300)
    err = p.stderr.splitlines()
    print("==== %s: rc %d" % (name, p.returncode))
    print("\n".join(p.stdout.splitlines()[-8:]))
    print("\n".join([l for l in err if l.startswith("[plan]")][-6:]))
    print("\n".join([l for l in err if not l.startswith("[plan]")][-12:]), flush=True)
