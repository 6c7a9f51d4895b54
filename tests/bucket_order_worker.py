"""Worker for test_dist_gpu.test_bucket_collectives_wait_for_both_gradient_streams (fresh process: the
stream -> hardware-queue mapping of a long-lived pytest process can serialise the two gradient streams and
hide the race).  One rank; the gradient-bucket collective is replaced by a snapshot taken with the
communicator's ordering rule (a third stream that waits on the CURRENT stream only).  Prints one JSON line.
usage: bucket_order_worker.py <main|side>   (which stream is slowed down with device sleeps)"""
import json, os, socket, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SEMSEG_FORCE_DIST"] = "1"
import torch
import torch.distributed as dist
from model.pspnet import PSPNet
from semseg_amd import trainer as T, engine as E, ops

lagging = sys.argv[1]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0)
torch.manual_seed(0)
m = PSPNet(layers=50, classes=7, zoom_factor=8, dropout=0.0, pretrained=False).cuda().train()
tr = T.Trainer(m, base_lr=0.01, sync_bn=True)
assert tr.dist_on
g = torch.Generator().manual_seed(11)
x = torch.randn(2, 3, 57, 57, generator=g).cuda()
y = torch.randint(0, 7, (2, 57, 57), generator=g).cuda()
coll = torch.cuda.Stream()
snaps, issued_from = [], []


class Work:
    def wait(self):
        torch.cuda.current_stream().wait_stream(coll)


def fake_all_reduce(t, op=None, group=None, async_op=False):
    if group is not tr.grad_group:
        return None                                   # SyncBN statistics: identity on one rank
    e = tr.engine(x)
    cur = torch.cuda.current_stream()
    issued_from.append("side" if (e._side is not None and cur == e._side) else "main")
    coll.wait_stream(cur)                             # the communicator's ordering rule
    with torch.cuda.stream(coll):
        snaps.append((t, t.clone()))
    return Work()


dist.all_reduce = fake_all_reduce
SLEEP = 20_000_000                                    # ~10 ms of device time per call
name = "conv_dgrad" if lagging == "main" else "conv_wgrad"
real = getattr(ops, name)


def slow(*a, **k):
    torch.cuda._sleep(SLEEP)
    return real(*a, **k)


setattr(ops, name, slow)


def run(steps=2):
    stale = None
    for _ in range(steps):
        snaps.clear(); issued_from.clear()
        tr.step(x, y)
        torch.cuda.synchronize()
        e = tr.engine(x)
        assert e._side is not None and len(snaps) == len(e._buckets) >= 2
        stale = [int((view != snap).sum().item()) for view, snap in snaps]
    return stale, list(issued_from)


stale_join, issued = run()
E.Engine.order_after_all_producers = lambda self: None          # negative control: the pre-fix behaviour
stale_ctl, _ = run()
print(json.dumps({"lagging": lagging, "issued_from": issued, "stale_with_join": stale_join,
                  "stale_without_join": stale_ctl}))
dist.destroy_process_group()
