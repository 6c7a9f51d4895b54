"""Which stream issues each gradient-bucket collective, and is the other stream still busy at that moment?
(1 rank, collective replaced by a recorder; weight-gradient kernels slowed down so the side stream lags.)"""
import os, sys, socket
sys.path.insert(0, ".")
os.environ["SEMSEG_FORCE_DIST"] = "1"
import torch, torch.distributed as dist
from model.pspnet import PSPNet
from semseg_amd import trainer as T, engine as E, ops
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0)
layers, size, batch = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (50, 57, 2)))
torch.manual_seed(0)
m = PSPNet(layers=layers, classes=7, zoom_factor=8, dropout=0.0, pretrained=False).cuda().train()
tr = T.Trainer(m, base_lr=0.01, sync_bn=True)
x = torch.randn(batch, 3, size, size).cuda(); y = torch.randint(0, 7, (batch, size, size)).cuda()
log = []
snaps = []
coll = torch.cuda.Stream()
real_ar = dist.all_reduce
def fake(t, op=None, group=None, async_op=False):
    if group is not tr.grad_group: return None
    e = tr.engine(x)
    cur = torch.cuda.current_stream()
    on_side = e._side is not None and cur == e._side
    other = e._main if on_side else e._side
    log.append(("side" if on_side else "main", None if other is None else (not other.query()), t.numel()))
    coll.wait_stream(cur)                     # communicator rule: ordered after the CURRENT stream only
    with torch.cuda.stream(coll):
        snaps.append((t, t.clone()))
    class W:
        def wait(self): torch.cuda.current_stream().wait_stream(coll)
    return W()
dist.all_reduce = fake
real_join = E.Engine.order_after_all_producers
join_first = len(sys.argv) > 4 and sys.argv[4] == 'join_first'
if not join_first:
    E.Engine.order_after_all_producers = lambda self: None      # observe the un-joined behaviour
real_w = ops.conv_wgrad
def slow_w(*a, **k):
    torch.cuda._sleep(20_000_000); return real_w(*a, **k)
ops.conv_wgrad = slow_w
for step in range(4):
    if join_first and step == 2:
        E.Engine.order_after_all_producers = lambda self: None; print('-- join disabled from here')
    log.clear(); snaps.clear(); tr.step(x, y); torch.cuda.synchronize()
    stale = [(i, int((v != c).sum().item()), v.numel()) for i, (v, c) in enumerate(snaps)]
    print("step %d: stale elements per bucket snapshot:" % step, stale, "| issue log:", [(a, b) for a, b, _ in log])
e = tr.engine(x)
print("buckets:", [(lo, hi, len(ps)) for lo, hi, ps in e._buckets])
names = {p: n for n, p in m.named_parameters()}
for bi, (lo, hi, ps) in enumerate(e._buckets):
    print(" bucket %d: first-listed %s ... last-listed %s" % (bi, names[ps[0]], names[ps[-1]]))
for i, (st, busy, n) in enumerate(log):
    print("collective %d issued from %s stream; other stream busy: %s; %d floats" % (i, st, busy, n))
dist.destroy_process_group()
