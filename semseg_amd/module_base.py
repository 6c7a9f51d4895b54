"""Shared nn.Module front-end of PSPNet / PSANet: engine cache + autograd bridge.

`forward(x, y=None)` keeps the reference contract (model/pspnet.py:80-105, model/psanet.py:154-179):
training -> (argmax int64 [N,h,w], main_loss, aux_loss) with autograd history, eval -> logits
[N,classes,h,w].  One custom autograd.Function spans the whole network: its forward runs the HIP
engine, its backward replays the engine tape and hands every parameter gradient back to autograd,
so `loss.backward()`, torch.optim.SGD, DistributedDataParallel and checkpointing work unchanged.
"""
import torch
from torch import nn

from .engine import Engine


def _holder_forward(self, *a, **k):
    raise RuntimeError("parameter container: executed by semseg_amd.engine on the MI355X, "
                       "call the enclosing PSPNet/PSANet instead")


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, y, ignore_index, *params):
        pred, main_loss, aux_loss = engine.forward_train(x, y, ignore_index)
        ctx.engine = engine
        ctx.mark_non_differentiable(pred)
        # 0-dim tensors, like nn.CrossEntropyLoss returns — fresh ones (the engine's own loss buffer is rewritten by the next
        # forward; `losses.append(main_loss)` in a caller's loop must keep every step's value, as with the reference).  `pred`
        # stays the engine's buffer (1.8 MB per image at 473 x 473): valid until the next forward of this engine, .clone() to keep.
        return pred, main_loss.clone().view(()), aux_loss.clone().view(())

    @staticmethod
    def backward(ctx, _gpred, gmain, gaux):
        eng = ctx.engine
        dev = eng.device
        zero = None
        if gmain is None or gaux is None:
            zero = torch.zeros(1, device=dev)
        gm = zero if gmain is None else gmain.reshape(1).contiguous().float()
        ga = zero if gaux is None else gaux.reshape(1).contiguous().float()
        eng.backward(gm, ga)
        grads = tuple(eng.grad_views[p] if p.requires_grad else None for p in eng.params)
        return (None, None, None, None) + grads


class HipSegModule(nn.Module):
    kind = "psp"

    MAX_ENGINES = 6   # each engine owns its activation buffers and packed weights: bound the cache (LRU)

    def _engine(self, x, training):
        if next(self.parameters(), None) is None:
            # nn.DataParallel replicas carry no nn.Parameters (tool/train.py's non-distributed multi-GPU path)
            raise RuntimeError("semseg_amd modules cannot run as multi-GPU nn.DataParallel replicas: use one process "
                               "per GPU with DistributedDataParallel (tool/train.py multiprocessing_distributed)")
        key = (tuple(x.shape), bool(training), x.device.index)
        cache = self.__dict__.setdefault("_engines", {})
        eng = cache.pop(key, None)
        if eng is None or eng.device != x.device or eng.params_stale():
            eng = Engine(self, x.shape[0], x.shape[2], x.shape[3], training, self.kind)
        cache[key] = eng                      # most recently used last
        while len(cache) > self.MAX_ENGINES:
            cache.pop(next(iter(cache)))      # evict the least recently used shape
        return eng

    def check_labels(self):
        """Blocks until the out-of-range-label counts of the training steps so far are on the host and raises IndexError if
        any is non-zero (see semseg_amd.engine.LabelWatch; eval forwards call this implicitly)."""
        for eng in self.__dict__.get("_engines", {}).values():
            eng.check_labels()
            break

    def _ignore_index(self):
        crit = getattr(self, "criterion", None)
        ii = getattr(crit, "ignore_index", 255)
        if not isinstance(crit, nn.CrossEntropyLoss) and crit is not None:
            raise NotImplementedError("the fused HIP head implements nn.CrossEntropyLoss(ignore_index) "
                                      "(tool/train.py:121); got %r" % (crit,))
        if isinstance(crit, nn.CrossEntropyLoss) and (crit.weight is not None or crit.reduction != "mean"
                                                      or getattr(crit, "label_smoothing", 0.0) != 0.0):
            raise NotImplementedError("only CrossEntropyLoss(ignore_index=..., reduction='mean')")
        return ii

    def forward(self, x, y=None):
        x_size = x.size()
        assert (x_size[2] - 1) % 8 == 0 and (x_size[3] - 1) % 8 == 0
        if not x.is_cuda:
            raise RuntimeError("semseg_amd runs on the MI355X only: move the model and input to cuda "
                               "(there is no CPU fallback)")
        if self.training:
            assert y is not None, "training forward needs the target (model/pspnet.py:101)"
            eng = self._engine(x, True)
            params = eng.params
            return _NetFunction.apply(eng, x.float(), y, self._ignore_index(), *params)
        eng = self._engine(x, False)
        with torch.no_grad():
            return eng.forward_eval(x.float())
