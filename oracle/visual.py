"""Oracle: CLIP ModifiedResNet image encoder, functional fp32 restatement.

Follows reference ``lib/models/backbones/m_resnet.py``:
  stem + layers  :146-217   (ModifiedResNet.__init__/forward)
  bottleneck     :14-67     (Bottleneck)
  attention pool :71-135    (AttentionPool2d)
State-dict names are the reference's (CLIP naming), tensors in reference
layout (conv OIHW, NCHW activations).  Test infrastructure only.
"""

from dataclasses import dataclass

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default used at m_resnet.py:19
BN_MOMENTUM = 0.1

# configs[3]'s "bf16" (BASELINE.json; the reference itself has no reduced-precision path: its DTYPE key is unread,
# lib/config/defaults.py:142-144): every convolution of the image encoder's 16 / 33 RESIDUAL BLOCKS (97 % of its
# FLOPs) takes its operands in bf16 - activations, filters and, in backward, the incoming gradient are rounded to bf16
# (round-to-nearest-even) where the convolution reads them - and every block OUTPUT is a bf16 tensor (so the identity
# residual of the next block reads the rounded value too); the gradient w.r.t. such a bf16 ACTIVATION tensor (a conv
# input, a conv output, a block output) is a bf16 tensor as well (as under autocast), and so is the raw conv OUTPUT (the
# BatchNorm statistics are those of the rounded tensor); products are exact, accumulation / BatchNorm arithmetic /
# weight gradients / the 3-conv stem / the attention pool's own arithmetic / everything else stays fp32.  `with bf16_conv():` switches
# the oracle to that arithmetic: the comparator of the HIP path's TRID_CONV_PRECISION=1 mode.
BF16_CONV = False


class bf16_conv:
    def __enter__(self):
        global BF16_CONV
        self.old, BF16_CONV = BF16_CONV, True

    def __exit__(self, *a):
        global BF16_CONV
        BF16_CONV = self.old


def _to_bf16(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _RoundOperand(torch.autograd.Function):
    """operand -> bf16 value (straight-through: the rounding is where the conv READS the tensor, not a layer)"""

    @staticmethod
    def forward(ctx, x):
        return _to_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundActivation(torch.autograd.Function):
    """activation -> bf16 value; the gradient w.r.t. a bf16 tensor is a bf16 tensor (rounded on its way back)"""

    @staticmethod
    def forward(ctx, x):
        return _to_bf16(x)

    @staticmethod
    def backward(ctx, g):
        return _to_bf16(g)


def _conv(x, w, round_input_grad=True, **kw):
    if BF16_CONV:
        # the conv OUTPUT is a bf16 tensor too (BatchNorm sees the rounded values); the gradient arriving at it - an
        # operand of the two backward convs - is rounded on its way in.  round_input_grad=False: the input is a block
        # input, whose gradient (this conv's share + the shortcut's) is rounded ONCE where that tensor was produced
        xin = _RoundActivation.apply(x) if round_input_grad else _RoundOperand.apply(x)
        return _RoundActivation.apply(F.conv2d(xin, _RoundOperand.apply(w), **kw))
    return F.conv2d(x, w, **kw)


@dataclass(frozen=True)
class VisualSpec:
    layers: tuple = (3, 4, 6, 3)
    width: int = 64
    heads: int = 32
    output_dim: int = 1024
    last_stride: int = 1
    height: int = 384
    in_width: int = 128

    @property
    def embed_dim(self):
        return self.width * 32  # m_resnet.py:182

    @property
    def spacial(self):
        r = 16 if self.last_stride == 1 else 32  # m_resnet.py:183-187
        return (self.height // r, self.in_width // r)


RN50 = VisualSpec()
RN101 = VisualSpec(layers=(3, 4, 23, 3), output_dim=512)
TINY = VisualSpec(layers=(1, 1, 1, 1), width=16, heads=4, output_dim=64, height=96, in_width=32)


def block_plan(spec):
    """[(prefix, inplanes, planes, stride)] in execution order (m_resnet.py:189-196)."""
    plan = []
    inpl = spec.width
    for li, (mult, nblk) in enumerate(zip((1, 2, 4, 8), spec.layers)):
        planes = spec.width * mult
        stride0 = 1 if li == 0 else (spec.last_stride if li == 3 else 2)
        for b in range(nblk):
            plan.append(("layer%d.%d" % (li + 1, b), inpl, planes, stride0 if b == 0 else 1))
            inpl = planes * 4
    return plan


def state_shapes(spec):
    """name -> shape for every parameter and buffer, reference order."""
    w = spec.width
    sh = {}

    def bn(p, c):
        sh[p + ".weight"] = (c,)
        sh[p + ".bias"] = (c,)
        sh[p + ".running_mean"] = (c,)
        sh[p + ".running_var"] = (c,)
        sh[p + ".num_batches_tracked"] = ()

    sh["conv1.weight"] = (w // 2, 3, 3, 3)
    bn("bn1", w // 2)
    sh["conv2.weight"] = (w // 2, w // 2, 3, 3)
    bn("bn2", w // 2)
    sh["conv3.weight"] = (w, w // 2, 3, 3)
    bn("bn3", w)
    for p, inpl, planes, stride in block_plan(spec):
        sh[p + ".conv1.weight"] = (planes, inpl, 1, 1)
        bn(p + ".bn1", planes)
        sh[p + ".conv2.weight"] = (planes, planes, 3, 3)
        bn(p + ".bn2", planes)
        sh[p + ".conv3.weight"] = (planes * 4, planes, 1, 1)
        bn(p + ".bn3", planes * 4)
        if stride > 1 or inpl != planes * 4:
            sh[p + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1)
            bn(p + ".downsample.1", planes * 4)
    e = spec.embed_dim
    hh, ww = spec.spacial
    sh["attnpool.positional_embedding"] = (hh * ww + 1, e)
    for nm, o in (("k_proj", e), ("q_proj", e), ("v_proj", e), ("c_proj", spec.output_dim)):
        sh["attnpool.%s.weight" % nm] = (o, e)
        sh["attnpool.%s.bias" % nm] = (o,)
    return sh


def is_param(name):
    leaf = name.split(".")[-1]
    return leaf not in ("running_mean", "running_var", "num_batches_tracked")


def _relu(x, taps):
    """F.relu; with a taps dict also tracks the smallest |pre-activation| seen ("relu_min"): the
    fixtures' ReLU decision margin (tests/golden/make_golden.py keeps it far above fp32 rounding).

    taps["force_masks"] (a list of bool tensors, consumed in call order): the ReLU DECISIONS are taken from the
    list instead of from the sign of x (relu(x) := x * mask).  A ReLU network is piecewise linear; two fp32
    evaluations with different summation orders disagree on the side of a few pre-activations that sit within
    rounding distance of zero, and past such a flip the two results are on different linear pieces.  Imposing the
    implementation-under-test's decisions puts the oracle on the SAME piece, where the comparison is a pure
    rounding-error comparison again; the disagreements themselves are counted ("flips") and their largest
    |pre-activation| recorded relative to the tensor's largest ("flip_max_rel") so a test can bound both."""
    if taps is not None:
        taps["relu_min"] = min(taps.get("relu_min", float("inf")), float(x.detach().abs().min()))
        if taps.get("force_masks"):
            m = taps["force_masks"].pop(0)
            xd = x.detach()
            dis = (xd > 0) != m
            n = int(dis.sum())
            taps["flips"] = taps.get("flips", 0) + n
            taps["relu_elems"] = taps.get("relu_elems", 0) + xd.numel()
            if n:
                taps["flip_max_rel"] = max(taps.get("flip_max_rel", 0.0), float(xd[dis].abs().max() / xd.abs().max()))
            return x * m.to(x.dtype)
    return F.relu(x)


def _bn(st, p, x, training):
    # nn.BatchNorm2d: batch stats + running update (momentum 0.1, unbiased var)
    # when training, running stats when not.  m_resnet.py:19,22,27,49,164-170
    rm, rv = st[p + ".running_mean"], st[p + ".running_var"]
    y = F.batch_norm(x, rm, rv, st[p + ".weight"], st[p + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training and (p + ".num_batches_tracked") in st:
        st[p + ".num_batches_tracked"] += 1
    return y


def bottleneck(st, p, x, stride, has_down, training, taps=None):
    # m_resnet.py:54-67
    out = _relu(_bn(st, p + ".bn1", _conv(x, st[p + ".conv1.weight"], round_input_grad=False), training), taps)
    out = _relu(_bn(st, p + ".bn2", _conv(out, st[p + ".conv2.weight"], padding=1), training), taps)
    if stride > 1:
        out = F.avg_pool2d(out, stride)
    out = _bn(st, p + ".bn3", _conv(out, st[p + ".conv3.weight"]), training)
    idn = x
    if has_down:
        if stride > 1:
            idn = F.avg_pool2d(idn, stride)
        idn = _bn(st, p + ".downsample.1", _conv(idn, st[p + ".downsample.0.weight"]), training)
    out = _relu(out + idn, taps)
    # bf16 mode: a block's OUTPUT tensor is a bf16 tensor (it is the next block's conv operand AND its identity
    # residual, the input of the downsample pooling, and - after the last block - of the attention pool)
    return _RoundActivation.apply(out) if BF16_CONV else out


def attention_pool(st, x, heads):
    """m_resnet.py:103-135.  The reference runs full MHA over all HW+1 tokens and
    returns token 0; mathematically only the query of token 0 is needed, which is
    what is computed here (checked against the imported reference in
    tests/golden/make_golden.py)."""
    B, C, H, W = x.shape
    tok = x.reshape(B, C, H * W).permute(0, 2, 1)  # [B,HW,C]
    tok = torch.cat([tok.mean(dim=1, keepdim=True), tok], dim=1)
    tok = tok + st["attnpool.positional_embedding"][None]
    hd = C // heads
    q = F.linear(tok[:, 0], st["attnpool.q_proj.weight"], st["attnpool.q_proj.bias"]) * (hd ** -0.5)
    k = F.linear(tok, st["attnpool.k_proj.weight"], st["attnpool.k_proj.bias"])
    v = F.linear(tok, st["attnpool.v_proj.weight"], st["attnpool.v_proj.bias"])
    q = q.view(B, heads, hd)
    k = k.view(B, -1, heads, hd)
    v = v.view(B, -1, heads, hd)
    s = torch.einsum("bhd,bthd->bht", q, k)
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bht,bthd->bhd", p, v).reshape(B, C)
    return F.linear(o, st["attnpool.c_proj.weight"], st["attnpool.c_proj.bias"])


def visual_forward(st, images, spec, training, taps=None):
    """images [B,3,H,W] f32 NCHW -> [B,output_dim].  ``st`` is mutated (BN
    running stats) when training, as nn.BatchNorm2d does.  ``taps`` (dict)
    optionally collects per-stage activations for the parity tests."""
    x = images.to(st["conv1.weight"].dtype)  # m_resnet.py:209
    x = _relu(_bn(st, "bn1", F.conv2d(x, st["conv1.weight"], stride=2, padding=1), training), taps)
    x = _relu(_bn(st, "bn2", F.conv2d(x, st["conv2.weight"], padding=1), training), taps)
    x = _relu(_bn(st, "bn3", F.conv2d(x, st["conv3.weight"], padding=1), training), taps)
    x = F.avg_pool2d(x, 2)
    if BF16_CONV:
        x = _RoundActivation.apply(x)  # the first block's input is a bf16 tensor like every other block's
    if taps is not None:
        taps["stem"] = x
    for p, inpl, planes, stride in block_plan(spec):
        has_down = stride > 1 or inpl != planes * 4
        x = bottleneck(st, p, x, stride, has_down, training, taps)
        if taps is not None:
            taps[p] = x
    out = attention_pool(st, x, spec.heads)
    if taps is not None:
        taps["attnpool"] = out
    return out
