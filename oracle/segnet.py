"""TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Functional restatement (plain torch CPU ops on a state_dict, no nn.Module tree) of the reference's
PSPNet / PSANet forward:
  trunk      /root/reference/model/resnet.py:74-94 (Bottleneck), :106-115 (deep stem + maxpool),
             :130-145 (_make_layer: first block of a stage has the projection shortcut)
  surgery    /root/reference/model/pspnet.py:49-58 (layer3/4: stride 1, dilation 2/4)
  PPM        /root/reference/model/pspnet.py:21-26
  PSA        /root/reference/model/psanet.py:53-98
  heads/loss /root/reference/model/pspnet.py:92-105, criterion from tool/train.py:121
The dense arithmetic (conv2d, batch_norm, interpolate, pooling, cross_entropy) is torch's CPU
implementation — the same third-party code the reference itself calls (README.md:11 pins
pytorch 1.4.0; this image has 2.10) — so parity of this oracle against the imported reference is
checked bit-for-bit by tests/golden/make_golden.py when the fixtures are generated.

Pinned by: tests/golden/*.npz (outputs of the imported reference, script committed).
"""
import torch
import torch.nn.functional as F

DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _bn(x, sd, p, training):
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    if training and p + ".num_batches_tracked" in sd:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training, 0.1, 1e-5)


def _block(x, sd, p, stride, dil, training):
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1", training))
    out = F.conv2d(out, sd[p + ".conv2.weight"], None, stride, dil, dil)
    out = F.relu(_bn(out, sd, p + ".bn2", training))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3", training)
    if p + ".downsample.0.weight" in sd:
        res = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride)
        res = _bn(res, sd, p + ".downsample.1", training)
    else:
        res = x
    return F.relu(out + res)


def trunk(sd, x, layers, training):
    x = F.relu(_bn(F.conv2d(x, sd["layer0.0.weight"], None, 2, 1), sd, "layer0.1", training))
    x = F.relu(_bn(F.conv2d(x, sd["layer0.3.weight"], None, 1, 1), sd, "layer0.4", training))
    x = F.relu(_bn(F.conv2d(x, sd["layer0.6.weight"], None, 1, 1), sd, "layer0.7", training))
    x = F.max_pool2d(x, 3, 2, 1)
    spec = [(1, 1), (2, 1), (1, 2), (1, 4)]  # (first-block stride, dilation) after the surgery
    x_tmp = None
    for li, (n, (stride, dil)) in enumerate(zip(DEPTHS[layers], spec)):
        for b in range(n):
            x = _block(x, sd, "layer%d.%d" % (li + 1, b), stride if b == 0 else 1, dil, training)
        if li == 2:
            x_tmp = x
    return x_tmp, x


def ppm(sd, x, bins, training):
    outs = [x]
    for i, b in enumerate(bins):
        f = F.adaptive_avg_pool2d(x, b)
        f = F.conv2d(f, sd["ppm.features.%d.1.weight" % i])
        f = F.relu(_bn(f, sd, "ppm.features.%d.2" % i, training))
        outs.append(F.interpolate(f, x.shape[2:], mode="bilinear", align_corners=True))
    return torch.cat(outs, 1)


def head(sd, x, p, training, dropmask=None):
    x = F.conv2d(x, sd[p + ".0.weight"], None, 1, 1)
    x = F.relu(_bn(x, sd, p + ".1", training))
    if dropmask is not None:  # Dropout2d(p) with an explicit per-(n,c) keep/scale mask
        x = x * dropmask[:, :, None, None]
    return F.conv2d(x, sd[p + ".4.weight"], sd[p + ".4.bias"])


def _perm(fn, t, *a):
    """The C oracle is fp32-only like the reference (psamask.cpp:117 `.data<float>()`); the op is a pure
    permutation, so an fp64 tensor is moved as a hi/lo pair of fp32 parts (exact to ~1e-15)."""
    if t.dtype == torch.float32:
        return torch.from_numpy(fn(t.detach().contiguous().numpy(), *a))
    hi = t.detach().float()
    lo = (t.detach() - hi.double()).float()
    return (torch.from_numpy(fn(hi.contiguous().numpy(), *a)).double() +
            torch.from_numpy(fn(lo.contiguous().numpy(), *a)).double())


class _PsaMask(torch.autograd.Function):
    """lib/psa/functions/psamask.py:6-39 on top of the C oracle (oracle/psamask_oracle.c)."""

    @staticmethod
    def forward(ctx, inp, psa_type, mH, mW):
        from . import psamask as pm
        ctx.cfg = (psa_type, mH, mW)
        return _perm(pm.psa_mask_forward, inp, psa_type, mH, mW)

    @staticmethod
    def backward(ctx, g):
        from . import psamask as pm
        t, mH, mW = ctx.cfg
        return _perm(pm.psa_mask_backward, g, t, mH, mW), None, None, None


def psa(sd, x, cfg, training):
    """model/psanet.py:53-98.  cfg: psa_type, compact, shrink_factor, mask_h, mask_w,
    normalization_factor, psa_softmax."""
    t, compact, sf = cfg["psa_type"], cfg["compact"], cfg["shrink_factor"]
    mh, mw = cfg["mask_h"], cfg["mask_w"]
    nf = cfg["normalization_factor"]
    if nf is None:
        nf = mh * mw
    out = x

    def reduce(p):
        return F.relu(_bn(F.conv2d(x, sd[p + ".0.weight"]), sd, p + ".1", training))

    def attention(z, p):
        z = F.relu(_bn(F.conv2d(z, sd[p + ".0.weight"]), sd, p + ".1", training))
        return F.conv2d(z, sd[p + ".3.weight"])

    def shrink(z):
        if sf == 1:
            return z
        h, w = (z.shape[2] - 1) // sf + 1, (z.shape[3] - 1) // sf + 1
        return F.interpolate(z, size=(h, w), mode="bilinear", align_corners=True)

    def branch(z, y, typ):
        n, c, h, w = z.shape
        if compact:
            if typ == 1:
                y = y.view(n, h * w, h * w).transpose(1, 2).reshape(n, h * w, h, w)
        else:
            y = _PsaMask.apply(y, typ, mh, mw)
        if cfg["psa_softmax"]:
            y = F.softmax(y, dim=1)
        return torch.bmm(z.reshape(n, c, h * w), y.reshape(n, h * w, h * w)).view(n, c, h, w) * (1.0 / nf)

    if t in (0, 1):
        z = shrink(reduce("psa.reduce"))
        z = branch(z, attention(z, "psa.attention"), t)
    else:
        zc, zd = shrink(reduce("psa.reduce")), shrink(reduce("psa.reduce_p"))
        z = torch.cat([branch(zc, attention(zc, "psa.attention"), 0),
                       branch(zd, attention(zd, "psa.attention_p"), 1)], 1)
    h, w = z.shape[2:]
    z = F.relu(_bn(F.conv2d(z, sd["psa.proj.0.weight"]), sd, "psa.proj.1", training))
    if sf != 1:
        z = F.interpolate(z, size=((h - 1) * sf + 1, (w - 1) * sf + 1), mode="bilinear", align_corners=True)
    return torch.cat((out, z), 1)


def forward(sd, x, layers, arch="psp", bins=(1, 2, 3, 6), zoom_factor=8, use_head=True, training=False,
            y=None, ignore_index=255, psa_cfg=None, dropmasks=None):
    """Returns logits (eval) or (argmax, main_loss, aux_loss) (training) like the reference forward."""
    H, W = x.shape[2:]
    assert (H - 1) % 8 == 0 and (W - 1) % 8 == 0
    h, w = int((H - 1) / 8 * zoom_factor + 1), int((W - 1) / 8 * zoom_factor + 1)
    x_tmp, f = trunk(sd, x, layers, training)
    if use_head:
        f = ppm(sd, f, bins, training) if arch == "psp" else psa(sd, f, psa_cfg, training)
    dm = dropmasks or {}
    z = head(sd, f, "cls", training, dm.get("cls"))
    if zoom_factor != 1:
        z = F.interpolate(z, size=(h, w), mode="bilinear", align_corners=True)
    if not training:
        return z
    aux = head(sd, x_tmp, "aux", training, dm.get("aux"))
    if zoom_factor != 1:
        aux = F.interpolate(aux, size=(h, w), mode="bilinear", align_corners=True)
    main_loss = F.cross_entropy(z, y, ignore_index=ignore_index)
    aux_loss = F.cross_entropy(aux, y, ignore_index=ignore_index)
    return z.max(1)[1], main_loss, aux_loss


def recipe_state_dict(model_sd_shapes, seed):
    """Seeded per-key tensors: conv ~ N(0, 2/fan_in) (keeps activations O(1) through 100 layers),
    BN gamma in [0.5,1.5], beta ~ 0.1 N, running_mean ~ 0.1 N, running_var in [0.5,1.5] so eval-mode
    BN is not the identity.  The same recipe runs on the GPU box (torch CPU RNG is deterministic)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in model_sd_shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[k] = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif k.endswith(".weight"):
            sd[k] = torch.rand(shape, generator=g) + 0.5
        else:
            sd[k] = torch.randn(shape, generator=g) * 0.1
    return sd
