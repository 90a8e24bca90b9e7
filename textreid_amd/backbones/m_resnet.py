"""CLIP ModifiedResNet image encoder on the MI355X kernel library.

Operator surface of the reference ``lib/models/backbones/m_resnet.py``
(``ModifiedResNet`` :138-217, ``Bottleneck`` :11-67, ``AttentionPool2d`` :70-135,
``modified_resnet50/101`` :246-291, ``build_m_resnet`` :294-307): same module
tree, parameter names, shapes and ``out_channels``, so reference / CLIP state
dicts load unchanged.  The torch ``nn.Conv2d`` / ``nn.BatchNorm2d`` /
``nn.Linear`` objects below are parameter holders only -- their ``forward`` is
never called.  The whole encoder is ONE ``autograd.Function`` whose forward and
backward are explicit sequences of HIP kernel launches (NHWC activations,
implicit-GEMM convolutions, BatchNorm statistics from the GEMM epilogue,
attention pool evaluated for the token-0 query only).
"""

import logging
import math
import os
from collections import OrderedDict

import torch
from torch import nn

from .. import ops


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.avgpool = nn.AvgPool2d(stride) if stride > 1 else nn.Identity()
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        self.stride = stride
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(
                OrderedDict(
                    [
                        ("-1", nn.AvgPool2d(stride)),
                        ("0", nn.Conv2d(inplanes, planes * 4, 1, stride=1, bias=False)),
                        ("1", nn.BatchNorm2d(planes * 4)),
                    ]
                )
            )
        assert stride in (1, 2), "AvgPool2d(stride) is implemented for stride 1 and 2"

    def forward(self, x):  # pragma: no cover - the encoder runs as one fused Function
        raise RuntimeError("Bottleneck is a parameter holder; call ModifiedResNet.forward")


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim, embed_dim, num_heads, output_dim=None):
        super().__init__()
        self.spacial_dim = spacial_dim
        self.proj_conv = None
        self.positional_embedding = nn.Parameter(
            torch.randn(spacial_dim[0] * spacial_dim[1] + 1, embed_dim) / embed_dim ** 0.5
        )
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim or embed_dim)
        self.num_heads = num_heads

    def forward(self, x):  # pragma: no cover
        raise RuntimeError("AttentionPool2d is a parameter holder; call ModifiedResNet.forward")


def _w3x3(conv):
    """[N,C,3,3] parameter -> [N, 9*C] tap-major/channel-minor view (a free view
    when the parameter is channels_last, which build/ctor arrange)."""
    w = conv.weight
    N, C = w.shape[0], w.shape[1]
    return w.permute(0, 2, 3, 1).contiguous().reshape(N, 9 * C)


def _g3x3(dw, N, C):
    """[N, 9*C] gradient -> [N,C,3,3]-shaped (channels_last strided) tensor."""
    return dw.view(N, 3, 3, C).permute(0, 3, 1, 2)


def _bn_coeffs(bn, partials, M, training, counters=None):
    """Per-layer BatchNorm coefficients.  Training also advances `num_batches_tracked`; with `counters`
    (a list) the increment is deferred so the caller can bump all 55 counters in one launch."""
    if training:
        st = ops.bn_finalize(partials, M, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        if counters is None:
            bn.num_batches_tracked += 1
        else:
            counters.append(bn.num_batches_tracked)
        return st
    return ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var)


import os as _os
_SERIAL_WGRAD = "1" in (_os.environ.get("TRID_SERIAL_WGRAD", "0"), _os.environ.get("TRID_SERIAL", "0"))  # experiment: weight gradients on the main stream
_SERIAL_ATTN_WGRAD = _os.environ.get("TRID_SERIAL_ATTN_WGRAD", "0") == "1"  # attention pool's weight gradients on the main stream (A/B runs)
_EARLY_WPT = _os.environ.get("TRID_EARLY_WPT", "1") != "0"  # data-gradient filter forms packed during the forward (0: at the head of backward, A/B runs)
_BATCH_WGRAD = int(_os.environ.get("TRID_WGRAD_BATCH", "1"))  # residual blocks whose weight gradients share one event (0: one event per weight gradient, A/B runs)


class _WgradStream:
    """Weight-gradient GEMMs are off the critical path of backward (nothing downstream in the
    chain consumes them), so they run on a side HIP stream underneath the dgrad GEMMs and the
    HBM-bound BatchNorm-backward passes of the main stream."""

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        self.side = _WgradStream._streams.setdefault(device, torch.cuda.Stream(device=device))
        self.side.wait_stream(self.main)
        self.pending = []
        self.blocks = 0

    _streams = {}

    def run(self, fn, *tensors, keep=()):
        """fn(*tensors) on the side stream, after everything enqueued so far on main.  `keep`: further tensors the
        side-stream kernels read through closures (the operands' amax scalars): their storage must not be recycled
        by the main stream's allocator before the side stream is done with it either."""
        if _SERIAL_WGRAD:
            return fn(*tensors)
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        for t in tensors + tuple(k for k in keep if k is not None):
            t.record_stream(self.side)  # keep the operands alive for the side stream
        with torch.cuda.stream(self.side):
            out = fn(*tensors)
        out.record_stream(self.main)
        return out

    def defer(self, G, key, fn, tensors, keep=(), post=None):
        """G[key] = post(fn(*tensors)), run on the side stream at the next flush().  Every run() costs the main stream an event
        record (a marker packet: ~7 us of bubble on the stream that carries the step's critical path - 50 per backward pass in the
        recorded step's timeline); a residual block's three or four weight gradients share ONE by waiting until the block's
        last data gradient is enqueued (they start a few hundred microseconds later, on a stream that is never the long one)."""
        if _SERIAL_WGRAD or not _BATCH_WGRAD:
            out = self.run(fn, *tensors, keep=keep)
            G[key] = post(out) if post is not None else out
            return
        self.pending.append((G, key, fn, tensors, keep, post))

    def flush(self, block_end=False):
        if block_end:  # (TRID_WGRAD_BATCH = n: every n-th block end flushes)
            self.blocks += 1
            if self.blocks % max(_BATCH_WGRAD, 1) != 0:
                return
        if not self.pending:
            return
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.side.wait_event(ev)
        for G, key, fn, tensors, keep, post in self.pending:
            for t in tensors + tuple(k for k in keep if k is not None):
                t.record_stream(self.side)
            with torch.cuda.stream(self.side):
                out = fn(*tensors)
            out.record_stream(self.main)
            G[key] = post(out) if post is not None else out
        self.pending = []

    def join(self):
        self.flush()
        self.main.wait_stream(self.side)


class ConvArith:
    """Conv arithmetic of one encoder pass.  With the fp16 two-plane split (ops.conv_precision() == 16) every GEMM
    operand travels with its largest magnitude as a device scalar: `WA` maps id(conv weight) -> max|w| (weight_amax),
    `slot()` hands out zeroed scalars for the amax side outputs of the BatchNorm kernels that produce activations."""

    def __init__(self, device, WA, PB=None):
        self.device, self.WA = device, WA
        # P: arithmetic of the stem convolutions - 16 (fp16 two-plane split, needs WA) or None (the library's default);
        # PB: arithmetic of the residual blocks' convolutions - the same, or 1: operands rounded to bf16 where the conv
        # reads them (configs[3]'s arithmetic, TRID_CONV_PRECISION=1; oracle: oracle.visual.bf16_conv)
        self.P = 16 if WA else None
        self.PB = self.P if PB is None else PB

    def slot(self):
        return ops.amax_slot(self.device) if self.WA else None

    def wam(self, conv):
        return self.WA.get(id(conv.weight))


def _conv_weights(mod):
    """The conv filters under `mod` in module order (the order of weight_amax's scalars).  The module tree is static: walked once
    (nn.Module.modules() cost 1.8 ms of host time per train step when walked on every pass)."""
    ent = mod.__dict__.get("_conv_weight_list")
    # (Parameter OBJECTS are stable under .to() / load_state_dict / optimizer steps; a caller that assigns new nn.Parameters
    # is caught by the identity of the first and last filter)
    if ent is None or ent[1].weight is not ent[0][0] or ent[2].weight is not ent[0][-1]:
        cm = [m for m in mod.modules() if isinstance(m, nn.Conv2d)]
        ent = ([m.weight for m in cm], cm[0], cm[-1])
        mod.__dict__["_conv_weight_list"] = ent
        mod.__dict__.pop("_block_conv_weight_list", None)
    return ent[0]


def weight_amax(mod):
    """{id(conv weight): device scalar max|w|} for every conv filter under `mod`, ONE multi-tensor launch (fp16-split
    arithmetic, ops.CONV_PRECISION == 16).  The pointer table is rebuilt only when a parameter moves."""
    convs = _conv_weights(mod)
    key = tuple(w.data_ptr() for w in convs)
    plan = getattr(mod, "_wamax_plan", None)
    if plan is None or plan[0] != key:
        dev = convs[0].device
        ptrs = torch.tensor(list(key), dtype=torch.int64, device=dev)
        sizes = torch.tensor([w.numel() for w in convs], dtype=torch.int64, device=dev)
        plan = (key, ptrs, sizes)
        mod._wamax_plan = plan
    out = torch.zeros(len(convs), dtype=torch.float32, device=convs[0].device)
    ops.call("trid_amax_multi_f32", ops._p(plan[1]), ops._p(plan[2]), len(convs), ops._p(out), ops.stream())
    WA = {id(w): out[i : i + 1] for i, w in enumerate(convs)}
    WA["all"] = out  # the scalars as one array, in module order of the conv filters (p16_weights)
    return WA


def p16_eligible(mod, mult=32):
    """The residual blocks run on pre-split (P16) operands when every block channel count is a multiple of the 32-wide
    K group - 64 for the bf16 form - (all CLIP ModifiedResNets at width 64; not the width-16 test encoders)."""
    cache = mod.__dict__.setdefault("_p16_ok", {})
    if mult not in cache:
        cache[mult] = all(isinstance(m, nn.Conv2d) is False or (m.in_channels % mult == 0 and m.out_channels % mult == 0)
                          for blk in mod.blocks() for m in blk.modules())
    return cache[mult]


def p16_weights(mod, WA, transposed, fmt=1, stem_only=False):
    """{id(conv weight): ops.P16} for every residual-block filter of `mod`, ONE multi-tensor launch: the forward
    operands [N][taps*C] (filters as stored: 3x3 in OHWI) or, transposed, the data-gradient operands [C][taps*N] with
    the taps reversed.  fmt 1: P16 (scaled by WA's amax scalars); fmt 2: plain bf16.  Destination buffers and the
    pointer table are persistent per module."""
    convs = _conv_weights(mod)  # weight_amax's order
    index = {id(w): i for i, w in enumerate(convs)}
    mine = mod.__dict__.get("_block_conv_weight_list")
    if mine is None:
        blocks = mod.blocks() if hasattr(mod, "blocks") else [mod]  # (a single Bottleneck: per-block parity tests)
        mine = [m.weight for blk in blocks for m in blk.modules() if isinstance(m, nn.Conv2d)]
        mod.__dict__["_block_conv_weight_list"] = mine
    if fmt == 1 and hasattr(mod, "blocks") and stem_p16_channels(mod):
        mine = [mod.conv2.weight, mod.conv3.weight] + mine  # the stem's 3x3 convolutions run on P16 operands too (stem_forward_p16)
    if stem_only:  # bf16 mode: the residual blocks read bf16 filters, the stem stays fp32-class on ITS two P16 filters
        mine = [mod.conv2.weight, mod.conv3.weight]
    key = tuple(w.data_ptr() for w in mine)
    plans = mod.__dict__.setdefault("_p16_plans", {})
    plan = plans.get((transposed, fmt, stem_only))
    if plan is None or plan[0] != key:
        dev = mine[0].device
        dsts, rows = [], []
        for w in mine:
            N, C, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
            if T > 1 and not w.is_contiguous(memory_format=torch.channels_last):
                raise RuntimeError("3x3 filters must live in channels_last (OHWI) memory")
            d = ops.p16_empty((C, T * N) if transposed else (N, T * C), w, fmt)
            dsts.append(d)
            rows.append([w.data_ptr(), d.data_ptr(), N, T, C, index[id(w)]])
        plan = (key, torch.tensor(rows, dtype=torch.int64, device=dev), dsts)
        plans[(transposed, fmt, stem_only)] = plan
    ops.call("trid_p16_pack_multi_f32", ops._p(plan[1]), ops._p(WA["all"]), len(mine), 1 if transposed else 0, fmt, ops.stream())
    return {id(w): ops.P16(d, WA[id(w)] if fmt == 1 else None, fmt) for w, d in zip(mine, plan[2])}


def _pre_mask(y, st):
    """bool [..., C]: the ReLU decision of relu(bn(y)) where only the pooled activation is kept, taken by the SAME
    kernel arithmetic as the forward / backward passes (parity tests only)."""
    return ops.bn_apply(y, st, relu=True) > 0


def stem_forward(mod, images, ar, training, nbt, masks=None):
    """Three conv -> BatchNorm -> ReLU units + 2x2 average pool (m_resnet.py:199-207).  Returns
    (x NHWC, its amax scalar, record for stem_backward).  masks (a list, parity tests): receives the three ReLU
    decision masks in execution order."""
    B = images.shape[0]
    P = ar.P
    col, Ho, Wo = ops.stem_im2col(images)
    c1 = mod.conv1.weight
    w1p = torch.zeros(c1.shape[0], col.shape[1], device=c1.device, dtype=c1.dtype)
    w1p[:, : c1[0].numel()] = c1.detach().reshape(c1.shape[0], -1)
    y1, p1 = ops.conv1x1(col, w1p, stats=True) if training else (ops.conv1x1(col, w1p), None)
    st1 = _bn_coeffs(mod.bn1, p1, y1.shape[0], training, nbt)
    y1 = y1.view(B, Ho, Wo, -1)
    a_a1 = ar.slot()
    a1 = ops.bn_apply(y1, st1, relu=True, amax=a_a1)
    w2 = _w3x3(mod.conv2)
    kw2 = dict(prec=P, aa=a_a1, ba=ar.wam(mod.conv2))
    y2, p2 = ops.conv3x3(a1, w2, stats=True, **kw2) if training else (ops.conv3x3(a1, w2, **kw2), None)
    st2 = _bn_coeffs(mod.bn2, p2, B * Ho * Wo, training, nbt)
    a_a2 = ar.slot()
    a2 = ops.bn_apply(y2, st2, relu=True, amax=a_a2)
    w3 = _w3x3(mod.conv3)
    kw3 = dict(prec=P, aa=a_a2, ba=ar.wam(mod.conv3))
    y3, p3 = ops.conv3x3(a2, w3, stats=True, **kw3) if training else (ops.conv3x3(a2, w3, **kw3), None)
    st3 = _bn_coeffs(mod.bn3, p3, B * Ho * Wo, training, nbt)
    ax = ar.slot()
    x = ops.bn_apply_pool2(y3, st3, relu=True, amax=ax)
    if masks is not None:
        masks.extend([a1 > 0, a2 > 0, _pre_mask(y3, st3)])
    return x, ax, (col, y1, st1, a1, y2, st2, a2, y3, st3, a_a2, a_a1)


def stem_backward(mod, rec, g, ar, ws, G):
    """Backward of stem_forward: fills G[id(param)] for the stem's nine parameters (the input image needs no gradient)."""
    col, y1, st1, a1, y2, st2, a2, y3, st3, a_a2, a_a1 = rec
    P = ar.P
    a_dy3 = ar.slot()
    dy3, dg, db, _ = ops.bn_bwd(g, y3, st3, None, 1, pooled=True, amax=a_dy3)
    G[id(mod.bn3.weight)], G[id(mod.bn3.bias)] = dg, db
    c3o, c3i = mod.conv3.out_channels, mod.conv3.in_channels
    w3t = ops.weight_transpose(_w3x3(mod.conv3), c3o, 9, c3i, flip=True)
    da2 = ops.conv3x3(dy3, w3t, prec=P, aa=a_dy3, ba=ar.wam(mod.conv3))
    G[id(mod.conv3.weight)] = _g3x3(ws.run(lambda d_, x_: ops.conv3x3_wgrad(d_, x_, prec=P, aa=a_dy3, ba=a_a2), dy3, a2, keep=(a_dy3, a_a2)), c3o, c3i)
    a_dy2 = ar.slot()
    dy2, dg, db, _ = ops.bn_bwd(da2, y2, st2, None, 1, amax=a_dy2)
    G[id(mod.bn2.weight)], G[id(mod.bn2.bias)] = dg, db
    c2o, c2i = mod.conv2.out_channels, mod.conv2.in_channels
    w2t = ops.weight_transpose(_w3x3(mod.conv2), c2o, 9, c2i, flip=True)
    da1 = ops.conv3x3(dy2, w2t, prec=P, aa=a_dy2, ba=ar.wam(mod.conv2))
    G[id(mod.conv2.weight)] = _g3x3(ws.run(ops.conv3x3_wgrad, dy2, a1), c2o, c2i)
    dy1, dg, db, _ = ops.bn_bwd(da1, y1, st1, None, 1)
    G[id(mod.bn1.weight)], G[id(mod.bn1.bias)] = dg, db
    dw1 = ws.run(ops.conv1x1_wgrad, dy1, col)  # [32, 28]
    c1 = mod.conv1.weight
    G[id(c1)] = dw1[:, : c1[0].numel()].reshape(c1.shape)


def stem_p16_channels(mod):
    """The P16 stem flow covers the CLIP stem's channel counts: 3 -> 32 -> 32 -> 64."""
    return (mod.conv1.in_channels, mod.conv1.out_channels, mod.conv2.out_channels, mod.conv3.out_channels) == (3, 32, 32, 64)


def p16_fits(images, width=64):
    """The P16 kernels address their operands with 31-bit buffer offsets: the largest activation of the pass - the stem's
    conv3 output and layer1's block outputs, B x H/2 x W/2 x `width` x 4 bytes - must stay below 2 GB (about 640 images of
    384 x 128 per GPU); larger per-GPU batches run the on-the-fly-split flow, whose kernels fall back per launch."""
    Ho, Wo = (images.shape[2] + 1) // 2, (images.shape[3] + 1) // 2
    return images.shape[0] * Ho * Wo * width * 4 < (1 << 31)


def stem_p16_ok(mod, images):
    """... and the map sizes the ring-of-rows convolution kernel tiles (csrc/stem_conv.hip): 192x64 at 384x128 input."""
    Ho, Wo = (images.shape[2] + 1) // 2, (images.shape[3] + 1) // 2
    return (stem_p16_channels(mod) and Ho % 2 == 0 and Wo % 2 == 0 and images.is_contiguous() and p16_fits(images, mod.conv3.out_channels)
            and all(ops.conv3x3_halo_rows(Ho, Wo, ci, co) > 0 for ci, co in ((32, 32), (32, 64), (64, 32))))


def stem_forward_p16(mod, images, WP, dev, nbt, masks=None, out_fmt=1):
    """The stem (m_resnet.py:199-207) as bandwidth-shaped kernels on pre-split operands: conv1 straight from the NCHW
    image (exact fp32 MFMA, no im2col tensor), conv2 / conv3 on the ring-of-rows kernel with their inputs written as
    P16 tensors by the BatchNorm passes (bounds from the conv epilogues' column extremes), and the pooled output handed
    to the residual blocks as the P16 tensor they consume (no fp32 copy + pack pass).  Returns (x P16, record)."""
    y1, p1 = ops.stem_conv1(images, mod.conv1.weight)
    M = y1.numel() // y1.shape[-1]
    b1 = ops.amax_slot(dev)
    st1 = _finalize_minmax(mod.bn1, p1, M, True, b1, nbt)
    a1 = ops.bn_apply_p16(y1, st1, b1, relu=True)
    y2, p2, rpp2 = ops.conv3x3_halo_p16(a1, WP[id(mod.conv2.weight)])
    b2 = ops.amax_slot(dev)
    st2 = _finalize_minmax(mod.bn2, p2, M, True, b2, nbt, rpp2)
    a2 = ops.bn_apply_p16(y2, st2, b2, relu=True)
    y3, p3, rpp3 = ops.conv3x3_halo_p16(a2, WP[id(mod.conv3.weight)])
    b3 = ops.amax_slot(dev)
    st3 = _finalize_minmax(mod.bn3, p3, M, True, b3, nbt, rpp3)
    # (an average never exceeds the maximum: b3 bounds the pooled tensor; out_fmt 2 - bf16 mode - hands the blocks a bf16 tensor)
    x = ops.bn_apply_pool2_p16(y3, st3, b3, relu=True, fmt=out_fmt)
    if masks is not None:
        masks.extend([a1.unpack() > 0, a2.unpack() > 0, _pre_mask(y3, st3)])
    return x, (images, y1, st1, a1, y2, st2, a2, y3, st3)


def stem_backward_p16(mod, rec, g, WPT, ws, G):
    """Backward of stem_forward_p16.  g: dL/d(pooled stem output) fp32.  BatchNorm-backward outputs are P16 tensors, the
    data gradients of conv3 / conv2 run on the ring-of-rows kernel with the rotated / transposed filters, the weight
    gradients on the transposing P16 kernel (side stream); conv1's weight gradient gathers the image once more, inside its
    kernel (exact fp32 MFMA, no im2col tensor)."""
    images, y1, st1, a1, y2, st2, a2, y3, st3 = rec
    Bi, H, W, _ = y3.shape

    def wgrad(dy, act, C):
        if ops.conv3x3_wgrad_halo_rows(H, W, C, dy.shape[-1]):  # every pixel staged once (csrc/stem_conv.hip)
            return ws.run(lambda d_, x_: ops.conv3x3_wgrad_halo_p16(ops.P16(d_, dy.amax), ops.P16(x_, act.amax)), dy.data, act.data,
                          keep=(dy.amax, act.amax))
        return ws.run(lambda d_, x_: ops.wgrad_p16(ops.P16(d_, dy.amax), ops.P16(x_, act.amax), conv=(H, W, C)), dy.data, act.data,
                      keep=(dy.amax, act.amax))

    dy3, dg, db, _ = ops.bn_bwd_p16(g, y3, st3, 1, pooled=True)
    G[id(mod.bn3.weight)], G[id(mod.bn3.bias)] = dg, db
    da2 = ops.conv3x3_halo_p16(dy3, WPT[id(mod.conv3.weight)], stats=False)
    G[id(mod.conv3.weight)] = _g3x3(wgrad(dy3, a2, 32), 64, 32)
    dy2, dg, db, _ = ops.bn_bwd_p16(da2, y2, st2, 1)
    G[id(mod.bn2.weight)], G[id(mod.bn2.bias)] = dg, db
    da1 = ops.conv3x3_halo_p16(dy2, WPT[id(mod.conv2.weight)], stats=False)
    G[id(mod.conv2.weight)] = _g3x3(wgrad(dy2, a1, 32), 32, 32)
    dy1, dg, db, _ = ops.bn_bwd(da1, y1, st1, None, 1)
    G[id(mod.bn1.weight)], G[id(mod.bn1.bias)] = dg, db
    G[id(mod.conv1.weight)] = ws.run(lambda d_, i_: ops.stem_conv1_wgrad(i_, d_), dy1, images)


def block_forward(blk, x, ax, ar, training, save, nbt, masks=None):
    """One Bottleneck (m_resnet.py:54-67) on NHWC activations.  x: block input, ax: its amax scalar (or None).
    Returns (out, amax scalar of out, record for block_backward or None).  masks (a list, parity tests): receives
    the block's three ReLU decision masks in execution order."""
    P = ar.PB
    stride = blk.stride
    wa = blk.conv1.weight.view(blk.conv1.out_channels, -1)
    kwa = dict(prec=P, aa=ax, ba=ar.wam(blk.conv1))
    ya, pa = ops.conv1x1(x, wa, stats=True, **kwa) if training else (ops.conv1x1(x, wa, **kwa), None)
    Ma = ya.numel() // ya.shape[-1]
    sta = _bn_coeffs(blk.bn1, pa, Ma, training, nbt)
    a_aa = ar.slot()
    aa = ops.bn_apply(ya, sta, relu=True, amax=a_aa)
    wb = _w3x3(blk.conv2)
    kwb = dict(prec=P, aa=a_aa, ba=ar.wam(blk.conv2))
    yb, pb = ops.conv3x3(aa, wb, stats=True, **kwb) if training else (ops.conv3x3(aa, wb, **kwb), None)
    stb = _bn_coeffs(blk.bn2, pb, Ma, training, nbt)
    a_ab = ar.slot()
    ab = ops.bn_apply_pool2(yb, stb, relu=True, amax=a_ab) if stride > 1 else ops.bn_apply(yb, stb, relu=True, amax=a_ab)
    wc = blk.conv3.weight.view(blk.conv3.out_channels, -1)
    kwc = dict(prec=P, aa=a_ab, ba=ar.wam(blk.conv3))
    yc, pc = ops.conv1x1(ab, wc, stats=True, **kwc) if training else (ops.conv1x1(ab, wc, **kwc), None)
    Mc = yc.numel() // yc.shape[-1]
    stc = _bn_coeffs(blk.bn3, pc, Mc, training, nbt)
    xd = yd = std = a_xd = None
    a_out = ar.slot()
    if blk.downsample is not None:
        a_xd = ar.slot() if stride > 1 else ax
        xd = ops.bn_apply_pool2(x, None, amax=a_xd) if stride > 1 else x
        wd = blk.downsample[1].weight.view(blk.downsample[1].out_channels, -1)
        kwd = dict(prec=P, aa=a_xd, ba=ar.wam(blk.downsample[1]))
        yd, pd = ops.conv1x1(xd, wd, stats=True, **kwd) if training else (ops.conv1x1(xd, wd, **kwd), None)
        std = _bn_coeffs(blk.downsample[2], pd, Mc, training, nbt)
        out = ops.bn_apply(yc, stc, relu=True, res=yd, res_st=std, want_mask=save, amax=a_out)
    else:
        out = ops.bn_apply(yc, stc, relu=True, res=x, want_mask=save, amax=a_out)
    rec = None
    if save:
        out, rmask = out  # 1-bit ReLU mask of the block output for the backward pass
        rec = (x, ya, sta, aa, yb, stb, ab, yc, stc, xd, yd, std, rmask, (ax, a_aa, a_ab, a_xd))
    if masks is not None:
        masks.extend([aa > 0, _pre_mask(yb, stb) if stride > 1 else ab > 0, out > 0])
    return out, a_out, rec


def block_backward(blk, rec, g, ar, ws, G):
    """Backward of block_forward.  g: dL/d(out).  Fills G[id(param)] for the block's parameters (weight gradients on
    the side stream `ws`) and returns dL/d(x)."""
    x, ya, sta, aa, yb, stb, ab, yc, stc, xd, yd, std, rmask, (ax, a_aa, a_ab, a_xd) = rec
    P = ar.PB
    stride = blk.stride
    has_down = blk.downsample is not None
    a_dyc = ar.slot()
    dyc, dg, db, dres = ops.bn_bwd(g, yc, stc, None, 3, act=rmask, want_dres=not has_down, amax=a_dyc)
    G[id(blk.bn3.weight)], G[id(blk.bn3.bias)] = dg, db
    if has_down:
        a_dyd = ar.slot()
        dyd, dg, db, _ = ops.bn_bwd(g, yd, std, None, 3, act=rmask, amax=a_dyd)
        G[id(blk.downsample[2].weight)], G[id(blk.downsample[2].bias)] = dg, db
    wc = blk.conv3.weight.view(blk.conv3.out_channels, -1)
    dab = ops.matmul_nn(dyc.view(-1, dyc.shape[-1]), wc, prec=P, aa=a_dyc, ba=ar.wam(blk.conv3)).view(ab.shape)
    G[id(blk.conv3.weight)] = ws.run(lambda d_, x_: ops.conv1x1_wgrad(d_, x_, prec=P, aa=a_dyc, ba=a_ab), dyc, ab, keep=(a_dyc, a_ab)).view_as(blk.conv3.weight)
    a_dyb = ar.slot()
    dyb, dg, db, _ = ops.bn_bwd(dab, yb, stb, None, 1, pooled=stride > 1, amax=a_dyb)
    G[id(blk.bn2.weight)], G[id(blk.bn2.bias)] = dg, db
    planes = blk.conv2.out_channels
    wbt = ops.weight_transpose(_w3x3(blk.conv2), planes, 9, planes, flip=True)
    daa = ops.conv3x3(dyb, wbt, prec=P, aa=a_dyb, ba=ar.wam(blk.conv2))
    G[id(blk.conv2.weight)] = _g3x3(ws.run(lambda d_, x_: ops.conv3x3_wgrad(d_, x_, prec=P, aa=a_dyb, ba=a_aa), dyb, aa, keep=(a_dyb, a_aa)), planes, planes)
    a_dya = ar.slot()
    dya, dg, db, _ = ops.bn_bwd(daa, ya, sta, None, 1, amax=a_dya)
    G[id(blk.bn1.weight)], G[id(blk.bn1.bias)] = dg, db
    wa = blk.conv1.weight.view(blk.conv1.out_channels, -1)
    if has_down:
        wd = blk.downsample[1].weight.view(blk.downsample[1].out_channels, -1)
        dxd = ops.matmul_nn(dyd.view(-1, dyd.shape[-1]), wd, prec=P, aa=a_dyd, ba=ar.wam(blk.downsample[1])).view(xd.shape)
        G[id(blk.downsample[1].weight)] = ws.run(lambda d_, x_: ops.conv1x1_wgrad(d_, x_, prec=P, aa=a_dyd, ba=a_xd), dyd, xd, keep=(a_dyd, a_xd)).view_as(blk.downsample[1].weight)
        dx = ops.avgpool2_bwd(dxd) if stride > 1 else dxd
    else:
        dx = dres
    ops.matmul_nn(dya.view(-1, dya.shape[-1]), wa, out=dx.view(-1, dx.shape[-1]), accumulate=True, prec=P, aa=a_dya, ba=ar.wam(blk.conv1))
    G[id(blk.conv1.weight)] = ws.run(lambda d_, x_: ops.conv1x1_wgrad(d_, x_, prec=P, aa=a_dya, ba=ax), dya, x, keep=(a_dya, ax)).view_as(blk.conv1.weight)
    return dx


def block_forward_p16(blk, x, WP, dev, training, save, nbt, masks=None):
    """block_forward on pre-split operands: x and every GEMM operand are ops.P16 tensors written by their producers
    (BatchNorm apply passes, whose output magnitude is known from the conv epilogue's column extremes before they
    run), the convolutions stage them with LDS-DMA (csrc/gemm_p16.hip).  Same arithmetic as the fp16-split path.
    With x.fmt == 2 the same data flow runs on plain bf16 tensors (configs[3]'s arithmetic; no scales / bounds)."""
    assert training
    stride = blk.stride
    fmt = x.fmt

    def fin(bn, parts, M, relu):
        """(BatchNorm coefficients, bound of max|act(bn(y))| or None)"""
        if fmt == 2:
            return _bn_coeffs(bn, parts, M, True, nbt), None
        bound = ops.amax_slot(dev)
        return _finalize_minmax(bn, parts, M, relu, bound, nbt), bound

    ya, pa = ops.conv_p16(x, WP[id(blk.conv1.weight)])
    Ma = ya.numel() // ya.shape[-1]
    sta, a_aa = fin(blk.bn1, pa, Ma, True)
    aa = ops.bn_apply_p16(ya, sta, a_aa, relu=True, fmt=fmt)
    yb, pb = ops.conv_p16(aa, WP[id(blk.conv2.weight)], conv3=True)
    stb, a_ab = fin(blk.bn2, pb, Ma, True)
    ab = ops.bn_apply_pool2_p16(yb, stb, a_ab, relu=True, fmt=fmt) if stride > 1 else ops.bn_apply_p16(yb, stb, a_ab, relu=True, fmt=fmt)
    Mc = ab.data.numel() // ab.shape[-1]
    xd = yd = std = None
    if blk.downsample is None and fmt == 1 and not save and ops.conv1x1_bn_res_ok(Mc, blk.conv3.out_channels, ab.shape[-1]):
        # identity blocks of layer1 / layer2 (K = 64 / 128) in the KEY encoder (nothing is kept for a backward pass): y =
        # conv3(ab) is 4 N bytes per row to write and read back against 4 K to read ab again - a statistics-only pass, then
        # conv3 + bn3 + identity + ReLU in one kernel that never stores y (layer1: 259 us instead of 307, layer2: 129 / 167).
        # With y kept (the query encoder: its BatchNorm backward needs it) the fused pass is no faster than GEMM + apply
        # (346 vs 326 us, 166 vs 174: tools/exp/fused_bench.py), so that side stays on the three-kernel form
        pc = ops.conv1x1_stats_p16(ab, WP[id(blk.conv3.weight)])
        stc, a_c = fin(blk.bn3, pc, Mc, False)
        out, rmask, yc = ops.conv1x1_bn_res_p16(ab, WP[id(blk.conv3.weight)], stc, a_c, x, relu=True, want_mask=save, keep_y=save)
        rec = (x, ya, sta, aa, yb, stb, ab, yc, stc, xd, yd, std, rmask) if save else None
        if masks is not None:
            masks.extend([aa.unpack() > 0, ab.unpack() > 0, out.unpack() > 0])
        return out, rec
    yc, pc = ops.conv_p16(ab, WP[id(blk.conv3.weight)])
    stc, a_c = fin(blk.bn3, pc, Mc, False)
    if blk.downsample is not None:
        xd = ops.bn_apply_pool2_p16(x, None, x.amax, fmt=fmt) if stride > 1 else x
        yd, pd = ops.conv_p16(xd, WP[id(blk.downsample[1].weight)])
        std, a_d = fin(blk.downsample[2], pd, Mc, False)
        out = ops.bn_apply_p16(yc, stc, a_c, relu=True, res=yd, res_st=std, bound_res=a_d, want_mask=save, fmt=fmt)
    else:
        out = ops.bn_apply_p16(yc, stc, a_c, relu=True, res=x, bound_res=x.amax, want_mask=save, fmt=fmt)
    rec = None
    if save:
        out, rmask = out
        rec = (x, ya, sta, aa, yb, stb, ab, yc, stc, xd, yd, std, rmask)
    if masks is not None:
        masks.extend([aa.unpack() > 0, (ops.bn_apply(yb.float(), stb, relu=True) if stride > 1 else ab.unpack()) > 0, out.unpack() > 0])
    return out, rec


def _finalize_minmax(bn, partials, M, relu, bound, counters, rows_per_part=ops.STATS_ROWS):
    st = ops.bn_finalize_minmax(partials, M, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, bound, rows_per_part=rows_per_part)
    counters.append(bn.num_batches_tracked)
    return st


_MASKED_ACC = _os.environ.get("TRID_MASKED_ACC", "1") != "0"


_BN3_FUSE = _os.environ.get("TRID_BN3_FUSE", "1") != "0"  # bn3's backward sums from the GEMM that produces the block-output gradient (0: A/B runs)
_BN3_FUSE_MIN_PLANES = int(_os.environ.get("TRID_BN3_FUSE_MIN_PLANES", "256"))  # (A/B runs: 512 = only where the product is on the tile kernel anyway)


def block_backward_p16(blk, rec, g, WPT, ws, G, g_sums=None, prev_rec=None, prev_blk=None):
    """Backward of block_forward_p16.  g: dL/d(out) fp32.  The BatchNorm-backward passes write their outputs as P16
    tensors (bound of max|dy| from the reduce pass); data gradients on gemm_p16, weight gradients on the transposing
    P16 kernel (side stream).
    g_sums: bn3's BatchNorm-backward sums (ops.BnBwdSums with this block's ReLU bits), formed by the GEMM that produced g - the
    block processed before this one; prev_rec / prev_blk: the block whose OUTPUT gradient this call produces (processed next).
    Returns (dL/dx, the BnBwdSums of that block's bn3 or None)."""
    x, ya, sta, aa, yb, stb, ab, yc, stc, xd, yd, std, rmask = rec
    stride = blk.stride
    has_down = blk.downsample is not None

    fmt = x.fmt

    def wgrad(weight, dy, act, conv=None, post=None):
        """G[weight] = the weight gradient, computed on the side stream at the end of the block (ws.flush below)."""
        if conv is not None and fmt == 1 and ops.conv3x3_wgrad_halo_rows(conv[0], conv[1], conv[2], dy.shape[-1]):  # layer1's conv2
            fn = lambda d_, x_: ops.conv3x3_wgrad_halo_p16(ops.P16(d_, dy.amax), ops.P16(x_, act.amax))
        else:
            fn = lambda d_, x_: ops.wgrad_p16(ops.P16(d_, dy.amax, fmt), ops.P16(x_, act.amax, fmt), conv=conv)
        ws.defer(G, id(weight), fn, (dy.data, act.data), keep=(dy.amax, act.amax), post=post if post is not None else (lambda o: o.view_as(weight)))

    # identity blocks: dL/dx = relu_mask * g + conv1's data gradient.  The masked copy of g is never written: conv1's data
    # gradient lands on g itself with the mask applied to the old values in its epilogue (gemm_p16 cmask) - g is dead after
    # bn3's backward passes.  (TRID_MASKED_ACC=0: bn_bwd writes the masked copy, for A/B runs)
    masked_acc = _MASKED_ACC and not has_down and g.is_contiguous()
    if has_down and ops.bn_bwd_dual_ok(g, yc, yd, fmt):
        # bn3 and the downsample branch's BatchNorm share g and the mask: one reduce pass and one apply pass for both
        dyc, dg, db, dyd, dg2, db2 = ops.bn_bwd_dual_p16(g, rmask, yc, stc, yd, std)
        G[id(blk.bn3.weight)], G[id(blk.bn3.bias)] = dg, db
        G[id(blk.downsample[2].weight)], G[id(blk.downsample[2].bias)] = dg2, db2
        dres = None
    else:
        dyc, dg, db, dres = ops.bn_bwd_p16(g, yc, stc, 3, act=rmask, want_dres=not has_down and not masked_acc, fmt=fmt,
                                           presummed=g_sums if (g_sums is not None and g_sums.y is yc) else None)
        G[id(blk.bn3.weight)], G[id(blk.bn3.bias)] = dg, db
        if has_down:
            dyd, dg, db, _ = ops.bn_bwd_p16(g, yd, std, 3, act=rmask, fmt=fmt)
            G[id(blk.downsample[2].weight)], G[id(blk.downsample[2].bias)] = dg, db
    planes = blk.conv2.out_channels
    Mc = dyc.data.numel() // dyc.shape[-1]
    dab = ops.empty(tuple(ab.shape), g, dtype=g.dtype)  # (bf16 mode: data gradients are bf16 tensors, like g)
    # bn2's / bn1's backward sums come out of the epilogue of the data-gradient GEMM that produces their gradient (the tile
    # kernel: planes >= 128, no pool in between) - the reduce pass over (gradient, saved conv output) is then not run
    # (no shape of the RN50 / RN101 blocks leaves the streaming kernel for these sums: conv3's data gradient has K = 4 planes >= 256
    # and N = planes, which gemm_p16_stream_rows does not cover - checked for ADVICE r05: the step has the same 977 launches with such shapes excluded)
    sums_b = ops.BnBwdSums(yb, stb) if (stride == 1 and g.dtype == torch.float32 and ops.bn_bwd_fusable(yb, Mc, planes, fmt)) else None
    ops.gemm_p16(dyc, WPT[id(blk.conv3.weight)], dab, Mc, planes, dyc.shape[-1], planes, bn_bwd=sums_b)
    wgrad(blk.conv3.weight, dyc, ab)
    dyb, dg, db, _ = ops.bn_bwd_p16(dab, yb, stb, 1, pooled=stride > 1, fmt=fmt, presummed=sums_b)
    G[id(blk.bn2.weight)], G[id(blk.bn2.bias)] = dg, db
    Bi, H, W, _ = yb.shape
    Ma = Bi * H * W
    sums_a = None
    if fmt == 1 and ops.USE_HALO_BLOCKS and ops.conv3x3_halo_rows(H, W, planes, planes):
        daa = ops.conv3x3_halo_p16(dyb, WPT[id(blk.conv2.weight)], stats=False)  # (layer1: 64 channels, the stem's ring-of-rows kernel)
    else:
        daa = ops.empty(tuple(aa.shape), g, dtype=g.dtype)
        sums_a = ops.BnBwdSums(ya, sta) if (g.dtype == torch.float32 and ops.bn_bwd_fusable(ya, Ma, planes, fmt)) else None
        ops.gemm_p16(dyb, WPT[id(blk.conv2.weight)], daa, Ma, planes, 9 * planes, planes, conv=(H, W, planes), bn_bwd=sums_a)
    wgrad(blk.conv2.weight, dyb, aa, conv=(H, W, planes), post=lambda o: _g3x3(o, planes, planes))
    dya, dg, db, _ = ops.bn_bwd_p16(daa, ya, sta, 1, fmt=fmt, presummed=sums_a)
    G[id(blk.bn1.weight)], G[id(blk.bn1.bias)] = dg, db
    cin = blk.conv1.in_channels
    if has_down:
        dxd = ops.empty(tuple(xd.shape), g, dtype=g.dtype)
        ops.gemm_p16(dyd, WPT[id(blk.downsample[1].weight)], dxd, Mc, cin, dyd.shape[-1], cin)
        wgrad(blk.downsample[1].weight, dyd, xd)
        dx = ops.avgpool2_bwd(dxd) if stride > 1 else dxd
    else:
        dx = g if masked_acc else dres
    # dx is the OUTPUT gradient of the block in front of this one: when that block is an identity block its bn3 backward takes
    # g = dx masked by ITS ReLU bits - the sums come out of this GEMM's epilogue (the tile kernel; its reduce pass over dx and the
    # saved conv3 output, 60-100 us per block of layer3 / layer4 re-reading 100-200 MB, is then not run).  K = planes >= 256 only:
    # below that this product runs on the streaming kernel, which a tile kernel with a 2-4 tile K loop does not beat by the
    # reduce pass's cost (layer3, K = 256: 64-68 us on the tile kernel against 59-62 us streaming, profiles/r04g_k256acc.txt)
    nxt = None
    if (_BN3_FUSE and prev_rec is not None and prev_blk is not None and prev_blk.downsample is None and fmt == 1 and planes >= _BN3_FUSE_MIN_PLANES
            and dx.dtype == torch.float32 and dx.is_contiguous() and prev_rec[12] is not None
            and ops.bn_bwd_fusable(prev_rec[7], Ma, cin, fmt) and tuple(prev_rec[7].shape) == tuple(dx.shape)):
        nxt = ops.BnBwdSums(prev_rec[7], prev_rec[8], mask=prev_rec[12])  # (yc, stc, rmask of the block in front)
    ops.gemm_p16(dya, WPT[id(blk.conv1.weight)], dx, Ma, cin, planes, cin, accumulate=True, cmask=rmask if masked_acc else None, bn_bwd=nxt)
    wgrad(blk.conv1.weight, dya, x)
    ws.flush(block_end=True)
    return dx, nxt


class _EncoderFn(torch.autograd.Function):
    """forward(images, module, save, *params) -> [B, out_dim]; grads for every parameter.
    `save` (activations kept for backward) is decided by the caller: inside forward() grad mode is
    always off and ctx.needs_input_grad ignores torch.no_grad()."""

    @staticmethod
    def forward(ctx, images, mod, save, *params):
        if not mod.training and not save and mod.fold_eval_bn and getattr(mod, "_debug_taps", None) is None:
            if ops.USE_EVAL_P16 and ops.USE_P16 and ops.conv_precision() == 16 and p16_eligible(mod, 32) and stem_p16_ok(mod, images):
                out, saved = mod._run_forward_eval_p16(images), None  # CLIP geometries: the P16 kernels with fused eval epilogues
            else:
                out, saved = mod._run_forward_folded(images), None   # other widths / arithmetic modes: folded filters, on-the-fly split
        else:
            if save and not mod.training:
                # eval-mode BatchNorm (running statistics) has a different backward (dy = g * scale, no batch terms);
                # only the train-mode backward is implemented - refuse rather than return batch-statistics gradients
                raise NotImplementedError("ModifiedResNet: gradients through an eval-mode (running-statistics) forward are not "
                                          "implemented; call under torch.no_grad() or in train() mode")
            out, saved = mod._run_forward(images, save)
        ctx.mod = mod
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, gout):
        mod, saved = ctx.mod, ctx.saved
        ctx.saved = None
        if saved is None:
            raise RuntimeError("backward through an encoder forward that did not save activations")
        grads = mod._run_backward(saved, gout.contiguous())
        return (None, None, None) + tuple(grads)


class ModifiedResNet(nn.Module):
    def __init__(self, layers, output_dim, heads, last_stride=1, input_resolution=(224, 224), width=64):
        super().__init__()
        self.output_dim = output_dim
        self.out_channels = output_dim
        self.fold_eval_bn = True  # eval + no_grad: BatchNorm folded into the conv weights, ReLU/residual in the GEMM epilogue
        self.grad_sync = None     # data parallel: a parallel.GradReducer; backward stages gradients into it per residual layer
        self.input_resolution = input_resolution
        self.conv1 = nn.Conv2d(3, width // 2, kernel_size=3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, kernel_size=3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        self.avgpool = nn.AvgPool2d(2)
        self.relu = nn.ReLU(inplace=True)
        self._inplanes = width
        self.layer1 = self._make_layer(width, layers[0])
        self.layer2 = self._make_layer(width * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(width * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(width * 8, layers[3], stride=last_stride)
        embed_dim = width * 32
        down_ratio = 16 if last_stride == 1 else 32
        spacial_dim = (input_resolution[0] // down_ratio, input_resolution[1] // down_ratio)
        self.attnpool = AttentionPool2d(spacial_dim, embed_dim, heads, output_dim)
        self._to_channels_last()

    def _make_layer(self, planes, blocks, stride=1):
        layers = [Bottleneck(self._inplanes, planes, stride)]
        self._inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self._inplanes, planes))
        return nn.Sequential(*layers)

    def _to_channels_last(self):
        # 3x3 filters live in OHWI memory (channels_last) so the kernels read them in place
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.in_channels % 4 == 0:
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)

    def blocks(self):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                yield blk

    # ------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("textreid_amd.ModifiedResNet runs on the HIP kernel library only (CUDA tensors); no CPU fallback")
        x = x.type(self.conv1.weight.dtype).contiguous()
        params = list(self.parameters())
        save = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _EncoderFn.apply(x, self, save, *params)

    def _eval_plan(self, device):
        """Everything of the eval-mode pass that depends on the parameters only, kept until a parameter or BatchNorm buffer is
        replaced or written (ops.parameter_generation(): the library's own writers go through raw pointers): the P16 filters
        (one pack launch), the running-statistics BatchNorm coefficients of the 55 / 106 layers and the output-bound
        coefficients (max_n |scale_n| ||w_n||_1, max_n |shift_n|) of every convolution (one launch)."""
        key = (device, ops.parameter_generation()) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        plan = getattr(self, "_eval_plan_cache", None)
        if plan is not None and plan[0] == key:
            return plan[1]
        WA = weight_amax(self)
        WP = p16_weights(self, WA, False, 1)
        pairs = [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]
        for blk in self.blocks():
            pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
            if blk.downsample is not None:
                pairs.append((blk.downsample[1], blk.downsample[2]))
        sts = {id(bn): _bn_coeffs(bn, None, 0, False) for _, bn in pairs}
        coef = ops.eval_bound_coefs([(conv.weight.detach(), sts[id(bn)].scale, sts[id(bn)].shift) for conv, bn in pairs], device)
        # (conv1 multiplies its fp32 filter as stored: no P16 copy)
        E = {id(conv.weight): (WP.get(id(conv.weight)), sts[id(bn)], coef[i]) for i, (conv, bn) in enumerate(pairs)}
        self._eval_plan_cache = (key, E, WA, coef)  # (WA / coef own the scalars the P16 filters and rows refer to)
        return E

    def _run_forward_eval_p16(self, images):
        """Eval mode (test_net.py / inference.py:14-26, head.py:178-183) on the pre-split (P16) kernels of the training pass:
        every convolution of the residual blocks is ONE launch - LDS-DMA tile kernel (csrc/gemm_p16.hip) or the streaming
        short-K kernel (csrc/gemm_stream.hip) - whose epilogue applies the running-statistics BatchNorm (scale, shift), adds
        the P16 identity / downsample branch, clamps, and writes the next operand as a P16 tensor.  The output's fp16 scale
        comes from an analytic bound (csrc/gemm_common.h EvalBound) on the TRUE maximum of the input, which each epilogue folds
        into a device scalar while it writes: no amax pass, no fp32 activation, no elementwise pass except the three 2x2
        average pools.  The stem (conv1 straight from the NCHW batch) and layer1's 64-channel 3x3 convolutions run on the
        bandwidth-shaped kernels of csrc/stem_conv.hip with the same fused epilogue."""
        E = self._eval_plan(images.device)
        log = getattr(self, "_debug_eval_bounds", None)  # tools/eval_bounds.py: (layer, P16 output) of every fused epilogue

        def note(name, t):
            if log is not None:
                log.append((name, t))
            return t

        _, st1, c1 = E[id(self.conv1.weight)]
        a1 = note("conv1", ops.stem_conv1_eval_p16(images, self.conv1.weight, st1, c1, ops.amax(images)))
        a2 = note("conv2", ops.conv_eval_p16(a1, *E[id(self.conv2.weight)], relu=True, conv3=True))  # (ring-of-rows kernel)
        wp3, st3, c3 = E[id(self.conv3.weight)]
        if ops.conv3x3_halo_eval_pool_ok(a2.shape[1], a2.shape[2], a2.shape[3], wp3.shape[0]):
            x = note("conv3+pool", ops.conv3x3_halo_eval_p16(a2, wp3, st3, c3, relu=True, pool=True))  # conv3 + bn3 + ReLU + AvgPool2d(2) in one kernel
        else:  # (other widths: BatchNorm + ReLU + pool as one pass over the raw output, scaled by the epilogue's exact extremes)
            y3, parts, rows = ops.conv3x3_halo_p16(a2, wp3)
            x = ops.bn_apply_pool2_p16(y3, st3, ops.bn_eval_bound(ops.Partials(parts, rows), st3, True), relu=True)
        for bi, blk in enumerate(self.blocks()):
            stride = blk.stride
            aa = note("block%d.conv1" % bi, ops.conv_eval_p16(x, *E[id(blk.conv1.weight)], relu=True))
            if stride > 1 and ops.conv_eval_pool_ok(aa.shape[1], aa.shape[2], blk.conv2.out_channels):
                ab = ops.conv_eval_p16(aa, *E[id(blk.conv2.weight)], relu=True, conv3=True, pool=True)  # conv2 + bn2 + ReLU + AvgPool2d(2)
            else:
                ab = ops.conv_eval_p16(aa, *E[id(blk.conv2.weight)], relu=True, conv3=True)  # (layer1: the ring-of-rows kernel)
                if stride > 1:
                    ab = ops.bn_apply_pool2_p16(ab, None, ab.amax)
            note("block%d.conv2" % bi, ab)
            ident = x
            if blk.downsample is not None:
                xd = ops.bn_apply_pool2_p16(x, None, x.amax) if stride > 1 else x
                ident = note("block%d.downsample" % bi, ops.conv_eval_p16(xd, *E[id(blk.downsample[1].weight)], relu=False))
            x = note("block%d.conv3" % bi, ops.conv_eval_p16(ab, *E[id(blk.conv3.weight)], relu=True, res=ident))
        feat, _ = self._attnpool_forward(x, False)
        return feat

    def _run_forward_folded(self, images):
        """Eval mode (test_net.py / inference.py:14-26): BatchNorm uses running statistics, so it is
        folded into the conv weights (w * gamma*invstd, bias = beta - mean*gamma*invstd) and the
        ReLU / residual add run in the GEMM epilogue - no elementwise pass except the 2x2 average
        pools.  Same values as the unfolded eval path up to fp32 rounding of the folded weights."""
        B = images.shape[0]
        col, Ho, Wo = ops.stem_im2col(images)
        # fp16-split conv arithmetic needs every operand's largest magnitude: one streaming pass per activation (there
        # is no BatchNorm kernel here to emit it on the side) - ~4 % of the pass, which it halves
        P = 16 if ops.conv_precision() == 16 else None
        am = ops.amax if P else (lambda t: None)
        # the folded filters (and their magnitudes) depend on the parameters only: kept across calls until a parameter
        # or BatchNorm buffer is replaced or modified in place (an inference run encodes thousands of batches)
        # (ops.parameter_generation(): the library's own writers - FusedAdam, EMA, running statistics - write through
        # raw pointers and do not bump torch's _version)
        key = (P, images.device, ops.parameter_generation()) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cache = getattr(self, "_folded_cache", None)
        fresh = cache is None or cache[0] != key
        if fresh:
            cache = (key, [])
            self._folded_cache = cache
        entries, pos = cache[1], [0]

        def folded(make_w2d, bn):
            """(folded filter, bias, max|filter|) of the next conv in call order"""
            if fresh:
                w_, b_ = ops.fold_bn(make_w2d(), _bn_coeffs(bn, None, 0, False))
                entries.append((w_, b_, am(w_)))
            e = entries[pos[0]]
            pos[0] += 1
            return e

        def c1x1(x_, wb, **kw):
            return ops.conv1x1(x_, wb[0], bias=wb[1], prec=P, aa=am(x_), ba=wb[2], **kw)

        def c3x3(x_, wb):
            return ops.conv3x3(x_, wb[0], bias=wb[1], relu=True, prec=P, aa=am(x_), ba=wb[2])

        def stem_w():
            c1 = self.conv1.weight
            w1p = torch.zeros(c1.shape[0], col.shape[1], device=c1.device, dtype=c1.dtype)
            w1p[:, : c1[0].numel()] = c1.detach().reshape(c1.shape[0], -1)
            return w1p

        a1 = c1x1(col, folded(stem_w, self.bn1), relu=True).view(B, Ho, Wo, -1)
        a2 = c3x3(a1, folded(lambda: _w3x3(self.conv2), self.bn2))
        x = ops.bn_apply_pool2(c3x3(a2, folded(lambda: _w3x3(self.conv3), self.bn3)), None)
        for blk in self.blocks():
            stride = blk.stride
            aa = c1x1(x, folded(lambda: blk.conv1.weight.view(blk.conv1.out_channels, -1), blk.bn1), relu=True)
            ab = c3x3(aa, folded(lambda: _w3x3(blk.conv2), blk.bn2))
            if stride > 1:
                ab = ops.bn_apply_pool2(ab, None)
            if blk.downsample is not None:
                xd = ops.bn_apply_pool2(x, None) if stride > 1 else x
                ident = c1x1(xd, folded(lambda: blk.downsample[1].weight.view(blk.downsample[1].out_channels, -1), blk.downsample[2]))
            else:
                ident = x
            x = c1x1(ab, folded(lambda: blk.conv3.weight.view(blk.conv3.out_channels, -1), blk.bn3), relu=True, residual=ident)
        feat, _ = self._attnpool_forward(x, False)
        return feat

    def _weight_amax(self):
        return weight_amax(self)

    def _run_forward(self, images, save):
        training = self.training
        B = images.shape[0]
        S = {"B": B} if save else None
        # fp16-split conv arithmetic: every GEMM operand comes with its largest magnitude as a device scalar
        cp = ops.conv_precision()
        ar = ConvArith(images.device, weight_amax(self) if cp in (16, 1) else {}, 1 if cp == 1 else None)
        nbt = []  # num_batches_tracked buffers, incremented together at the end of the pass
        # ---- stem (m_resnet.py:199-207)
        masks = getattr(self, "_debug_masks", None)  # parity tests: every ReLU decision of the pass, in execution order
        p16 = ops.USE_P16 and training and ar.PB in (16, 1) and p16_eligible(self, 32 if ar.PB == 16 else 64) and p16_fits(images, self.conv3.out_channels)
        fmt = (1 if ar.PB == 16 else 2) if p16 else 0
        WP = p16_weights(self, ar.WA, False, fmt) if p16 else None  # every filter the pass multiplies with, ONE launch
        # the stem's bandwidth-shaped kernels are fp32-class (P16 operands, exact fp32 conv1) in BOTH modes: in the bf16 mode
        # (configs[3]) the stem, like BatchNorm and the weight gradients, keeps fp32 arithmetic and hands over a bf16 tensor
        stem16 = fmt in (1, 2) and ops.USE_P16_STEM and stem_p16_ok(self, images)
        if stem16:
            WPs = WP if fmt == 1 else p16_weights(self, ar.WA, False, 1, stem_only=True)
            x, srec = stem_forward_p16(self, images, WPs, images.device, nbt, masks, out_fmt=fmt)
        else:
            x, ax, srec = stem_forward(self, images, ar, training, nbt, masks)
        if save:
            S["stem"] = srec
            S["stem_p16"] = stem16
            S["wamax"] = ar.WA
            S["prec"] = ar.PB
            if p16 and _EARLY_WPT:
                # the data-gradient forms of the filters (taps reversed, transposed) are packed HERE, where the step is
                # throughput-bound, instead of at the head of backward, where the pack (130 us) sat alone on the critical path
                # between the attention pool's backward and layer4's (the weights do not change in between)
                S["WPT"] = p16_weights(self, ar.WA, True, fmt)
        # ---- residual layers (m_resnet.py:54-67)
        if save:
            S["blocks"] = []
        taps = getattr(self, "_debug_taps", None)  # parity tests: per-stage activations (NHWC), keyed like the oracle's taps
        if taps is not None:
            taps["stem"] = x.unpack() if stem16 else x
            names = {id(blk): "layer%d.%d" % (li + 1, bi) for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4))
                     for bi, blk in enumerate(layer)}
        if p16:
            # residual blocks on pre-split operands: filters packed once per pass, activations written split by the
            # BatchNorm passes.  The stem hands over a P16 tensor (stem_forward_p16) or, on geometries / modes its
            # kernels do not cover, an fp32 tensor that is packed here.  PB == 1 (configs[3]): the same data flow on
            # plain bf16 tensors
            if not stem16:
                x = ops.p16_pack(x, ax, fmt)
            for blk in self.blocks():
                x, rec = block_forward_p16(blk, x, WP, images.device, training, save, nbt, masks)
                if save:
                    S["blocks"].append(rec)
                if taps is not None:
                    taps[names[id(blk)]] = x.unpack()
            if save:
                S["p16"] = fmt  # (x stays in its P16 / bf16 form: the attention pool's token kernel decodes it)
        else:
            for blk in self.blocks():
                x, ax, rec = block_forward(blk, x, ax, ar, training, save, nbt, masks)
                if save:
                    S["blocks"].append(rec)
                if taps is not None:
                    taps[names[id(blk)]] = x
        if nbt:
            torch._foreach_add_(nbt, 1)  # one launch instead of one per BatchNorm layer
        # ---- attention pool (m_resnet.py:103-135), token-0 query only
        feat, asave = self._attnpool_forward(x, save)
        if save:
            S["attn"] = asave
        return feat, S

    def _attnpool_forward(self, x, save):
        ap = self.attnpool
        B, H, W, C = x.shape
        T = H * W
        T1 = T + 1
        T1p = (T1 + 3) // 4 * 4
        heads = ap.num_heads
        hd = C // heads
        scale = float(hd) ** -0.5
        xd = x.data if isinstance(x, ops.P16) else x
        x_fmt = x.fmt if isinstance(x, ops.P16) else 0
        tok = ops.empty((B, T1p, C), xd)
        ops.call("trid_attnpool_tokens_fmt_f32", ops._p(xd), x_fmt, ops._p(x.amax) if x_fmt == 1 else None, ops._p(ap.positional_embedding),
                 ops._p(tok), B, T, C, T1p, ops.stream())
        x = tok  # (shape / device carrier for what follows)
        q = ops.linear(tok[:, 0], ap.q_proj.weight, ap.q_proj.bias)  # [B,C]
        # U[b,h,:] = scale * q[b,h,:] @ Wk[h]  (k bias is softmax-invariant and dropped)
        U = ops.empty((B, heads, C), x)
        ops.gemm(q, ap.k_proj.weight, U, B, C, hd, C, C, heads * C, b_mode=ops.B_NC, alpha=scale, batch=heads,
                 strideA=hd, strideB=hd * C, strideC=C)
        # S[b,h,t] = U[b,h,:] . tok[b,t,:]
        P = ops.empty((B, heads, T1p), x)
        ops.gemm(U, tok, P, heads, T1p, C, C, C, T1p, batch=B, strideA=heads * C, strideB=T1p * C, strideC=heads * T1p)
        ops.softmax_rows_(P, T1)
        # Z[b,h,:] = sum_t P[b,h,t] tok[b,t,:]
        Z = ops.empty((B, heads, C), x)
        ops.gemm(P, tok, Z, heads, C, T1p, T1p, C, C, b_mode=ops.B_NC, batch=B, strideA=heads * T1p, strideB=T1p * C,
                 strideC=heads * C)
        # o[b, h*hd+d] = Wv[h*hd+d,:] . Z[b,h,:] + bv
        o = ops.empty((B, C), x)
        ops.gemm(Z, ap.v_proj.weight, o, B, hd, C, heads * C, C, C, batch=heads, strideA=C, strideB=hd * C, strideC=hd,
                 bias=ap.v_proj.bias, strideBias=hd)
        out = ops.linear(o, ap.c_proj.weight, ap.c_proj.bias)
        return out, (((B, H, W, C), tok, q, U, P, Z, o) if save else None)

    # ------------------------------------------------------------------ backward
    def _attnpool_backward(self, asave, gout, G, ws=None):
        ap = self.attnpool
        xshape, tok, q, U, P, Z, o = asave
        B, H, W, C = xshape
        T = H * W
        T1 = T + 1
        T1p = tok.shape[1]
        heads = ap.num_heads
        hd = C // heads
        scale = float(hd) ** -0.5
        Wq, Wk, Wv, Wc = ap.q_proj.weight, ap.k_proj.weight, ap.v_proj.weight, ap.c_proj.weight
        # c_proj
        # The four projection-weight gradients (and the bias sums) feed nothing downstream: like the convolutions' weight
        # gradients they run on the side stream, behind ONE event at the end of this function - on the main stream they sat
        # between the losses and layer4's backward, alone on the step's critical path (~0.25 ms: profiles/r05g_attn_wgrad_ab.txt)
        def side(key, fn, *tensors):
            if ws is None or _SERIAL_ATTN_WGRAD:
                G[key] = fn(*tensors)
            else:
                ws.defer(G, key, fn, tensors)

        do = ops.matmul_nn(gout, Wc)  # [B,C]
        side(id(Wc), lambda g_, o_: ops.matmul_tn(g_, o_), gout, o)
        side(id(ap.c_proj.bias), lambda g_: ops.colsum(g_), gout)
        side(id(ap.v_proj.bias), lambda d_: ops.colsum(d_), do)  # sum_t P = 1

        # dWv[h*hd+d, c] = sum_b do[b,h*hd+d] Z[b,h,c]
        def dwv(do_, Z_):
            out = torch.empty_like(Wv)
            ops.gemm(do_, Z_, out, hd, C, B, C, heads * C, C, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=heads, strideA=hd,
                     strideB=C, strideC=hd * C)
            return out

        side(id(Wv), dwv, do, Z)
        # dZ[b,h,c] = sum_d do[b,h*hd+d] Wv[h*hd+d,c]
        dZ = ops.empty((B, heads, C), tok)
        ops.gemm(do, Wv, dZ, B, C, hd, C, C, heads * C, b_mode=ops.B_NC, batch=heads, strideA=hd, strideB=hd * C,
                 strideC=C)
        # dP[b,h,t] = dZ[b,h,:] . tok[b,t,:]
        dP = ops.empty((B, heads, T1p), tok)
        ops.gemm(dZ, tok, dP, heads, T1p, C, C, C, T1p, batch=B, strideA=heads * C, strideB=T1p * C,
                 strideC=heads * T1p)
        dS = ops.softmax_rows_bwd(P, dP, T1, out=dP)
        # dtok[b,t,:] = sum_h P[b,h,t] dZ[b,h,:] + dS[b,h,t] U[b,h,:]
        dtok = ops.empty((B, T1p, C), tok)
        ops.gemm(P, dZ, dtok, T1p, C, heads, T1p, C, C, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=B,
                 strideA=heads * T1p, strideB=heads * C, strideC=T1p * C)
        ops.gemm(dS, U, dtok, T1p, C, heads, T1p, C, C, a_mode=ops.A_MC, b_mode=ops.B_NC, batch=B,
                 strideA=heads * T1p, strideB=heads * C, strideC=T1p * C, accumulate=True)
        # dU[b,h,:] = sum_t dS[b,h,t] tok[b,t,:]
        dU = ops.empty((B, heads, C), tok)
        ops.gemm(dS, tok, dU, heads, C, T1p, T1p, C, C, b_mode=ops.B_NC, batch=B, strideA=heads * T1p,
                 strideB=T1p * C, strideC=heads * C)
        # dq[b,h*hd+d] = scale * dU[b,h,:] . Wk[h*hd+d,:]
        dq = ops.empty((B, C), tok)
        ops.gemm(dU, Wk, dq, B, hd, C, heads * C, C, C, alpha=scale, batch=heads, strideA=C, strideB=hd * C, strideC=hd)
        # dWk[h*hd+d, c] = scale * sum_b q[b,h*hd+d] dU[b,h,c]
        def dwk(q_, dU_):
            out = torch.empty_like(Wk)
            ops.gemm(q_, dU_, out, hd, C, B, C, heads * C, C, a_mode=ops.A_MC, b_mode=ops.B_NC, alpha=scale, batch=heads,
                     strideA=hd, strideB=C, strideC=hd * C)
            return out

        side(id(Wk), dwk, q, dU)
        G[id(ap.k_proj.bias)] = torch.zeros_like(ap.k_proj.bias)  # softmax is shift-invariant: exactly zero
        # q projection (token 0)
        ops.matmul_nn(dq, Wq, out=dtok[:, 0], accumulate=True)  # (rows of dtok at pitch T1p * C: token 0 of every image)
        side(id(Wq), lambda dq_, tok_: ops.matmul_tn(dq_, tok_[:, 0]), dq, tok)
        side(id(ap.q_proj.bias), lambda dq_: ops.colsum(dq_), dq)
        dx = ops.empty((B, H, W, C), tok)
        dpos = torch.empty_like(ap.positional_embedding)
        ops.call("trid_attnpool_tokens_bwd_f32", ops._p(dtok), ops._p(dx), ops._p(dpos), B, T, C, T1p, ops.stream())
        G[id(ap.positional_embedding)] = dpos
        if ws is not None:
            ws.flush()
        return dx

    def _run_backward(self, S, gout):
        G = {}
        ws = _WgradStream(gout.device)
        ar = ConvArith(gout.device, S.get("wamax", {}), S.get("prec"))  # the forward's conv arithmetic (its weight / activation amax scalars are reused here)
        g = self._attnpool_backward(S["attn"], gout, G, ws)
        S["attn"] = None
        blocks = list(self.blocks())
        sync = self.grad_sync
        staged = set()

        def stage_ready():
            """Data parallel: everything in G that has not been handed over yet is final - start its
            all-reduce now (after the weight-gradient stream has caught up), under the rest of backward."""
            if sync is None:
                return
            ws.join()
            ps = [p for p in self.parameters() if id(p) in G and id(p) not in staged and G[id(p)] is not None]
            staged.update(id(p) for p in ps)
            sync.stage(ps, [G[id(p)] for p in ps])

        first_of_layer = {id(layer[0]) for layer in (self.layer1, self.layer2, self.layer3, self.layer4)}
        dbg = getattr(self, "_debug_grads", None)
        p16 = S.get("p16", 0)
        WPT = (S.get("WPT") or p16_weights(self, ar.WA, True, p16)) if p16 else None
        if p16 == 2 and ops.BF16_GRADS:
            g = g.to(torch.bfloat16)  # bf16 mode: the gradient of a bf16 tensor (the block outputs) is a bf16 tensor
        rblocks, rrecs = list(reversed(blocks)), list(reversed(S["blocks"]))
        g_sums = None
        for bi, (blk, rec) in enumerate(zip(rblocks, rrecs)):
            if dbg is not None:
                dbg.append(g.float().clone())  # (a copy: identity blocks overwrite g with dL/dx)
            if p16:
                more = bi + 1 < len(rblocks)
                g, g_sums = block_backward_p16(blk, rec, g, WPT, ws, G, g_sums=g_sums, prev_rec=rrecs[bi + 1] if more else None,
                                               prev_blk=rblocks[bi + 1] if more else None)
            else:
                g = block_backward(blk, rec, g, ar, ws, G)
            if id(blk) in first_of_layer:  # a whole residual layer (and, the first time, the attention pool) is done
                stage_ready()
        S["blocks"] = None
        g = g.float()  # the stem's BatchNorm passes take fp32 gradients in every mode
        if S.get("stem_p16"):
            stem_backward_p16(self, S["stem"], g, WPT if p16 == 1 else p16_weights(self, ar.WA, True, 1, stem_only=True), ws, G)
        else:
            stem_backward(self, S["stem"], g, ar, ws, G)
        ws.join()
        if sync is not None:
            stage_ready()  # the stem
            G.update(sync.finish_stages())
        return [G.get(id(p)) for p in self.parameters()]


def resize_pos_embed(posemb, gs_new):
    """CLIP 7x7 positional grid -> target grid, bilinear (m_resnet.py:220-233).
    Host-side checkpoint ingestion (torch interpolate = plumbing, runs once)."""
    import torch.nn.functional as F

    logging.getLogger("PersonSearch.train").info("Resized position embedding to %s", (gs_new,))
    tok, grid = posemb[:1], posemb[1:]
    gs_old = int(math.sqrt(len(grid)))
    grid = grid.reshape(1, gs_old, gs_old, -1).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, size=gs_new, mode="bilinear", align_corners=False)
    grid = grid.permute(0, 2, 3, 1).reshape(gs_new[0] * gs_new[1], -1)
    return torch.cat([tok, grid], dim=0)


def state_filter(state_dict, final_stage_resolution):
    out = {}
    for k, v in state_dict.items():
        if k.startswith("visual."):
            k = k[7:]
        if k == "attnpool.positional_embedding" and final_stage_resolution != (7, 7):
            v = resize_pos_embed(v, final_stage_resolution)
        out[k] = v
    return out


def _load_clip(model, pretrained_path):
    if pretrained_path and os.path.exists(pretrained_path):
        p = torch.jit.load(pretrained_path, map_location="cpu").state_dict()
        model.load_state_dict(state_filter(p, model.attnpool.spacial_dim), strict=False)
    elif pretrained_path:
        logging.getLogger("PersonSearch.train").warning("CLIP weights %s not found: random init", pretrained_path)
    return model


def modified_resnet50(input_resolution, last_stride, pretrained_path=None):
    m = ModifiedResNet([3, 4, 6, 3], 1024, 32, last_stride, input_resolution)
    return _load_clip(m, pretrained_path)


def modified_resnet101(input_resolution, last_stride, pretrained_path=None):
    m = ModifiedResNet([3, 4, 23, 3], 512, 32, last_stride, input_resolution)
    return _load_clip(m, pretrained_path)


def build_m_resnet(cfg):
    res = (cfg.INPUT.HEIGHT, cfg.INPUT.WIDTH)
    if cfg.MODEL.VISUAL_MODEL in ["m_resnet50", "m_resnet"]:
        return modified_resnet50(res, cfg.MODEL.RESNET.RES5_STRIDE, os.path.join(cfg.ROOT, "pretrained/clip/RN50.pt"))
    if cfg.MODEL.VISUAL_MODEL == "m_resnet101":
        return modified_resnet101(res, cfg.MODEL.RESNET.RES5_STRIDE, os.path.join(cfg.ROOT, "pretrained/clip/RN101.pt"))
    raise NotImplementedError(cfg.MODEL.VISUAL_MODEL)
