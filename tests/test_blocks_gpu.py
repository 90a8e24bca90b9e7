"""GPU parity at the BENCHMARKED shapes with UNSTRUCTURED ReLU masks.

The full-size fixtures use the `margin` fill (every ReLU all-on or all-off per channel) because a deep He-style
network's gradients are discontinuous at fp32 resolution.  That leaves the 1-bit ReLU mask path, `mask_mode 3` and the
fp16-split GEMM variants at bench tile shapes seeing only trivial masks.  Here ONE Bottleneck (three ReLUs: flips
stay countable) and the stem run at B = 128 on every distinct RN50 block shape with He-style weights and post-ReLU
style inputs - random, unstructured masks - through the model's own code path (`block_forward` / `block_backward`,
fp16 two-plane split convolutions with producer-side amax scalars, weight gradients on the side stream), against the
CPU oracle (`oracle/visual.py:bottleneck`, reference `m_resnet.py:54-67,198-217`) evaluated in fp64.

At M = B*H*W = 393 216 pixels a ReLU sees 25 M pre-activations; ~1e-6 of them lie within fp32 rounding distance of
zero, where two correct evaluations with different summation orders decide differently, and ONE flipped decision moves
a bias-like gradient (a random-sign sum over M pixels) by ~1/sqrt(M) = 1.6e-3 of its value - the oracle's own fp32
evaluation deviates from its fp64 evaluation by that much (measured on the first run of this test: layer1.0 dx max
3.9e-2, bias gradients 1-2e-3).  So the comparison is split into the two statements that ARE sharp:

  1. decisions: the HIP path's ReLU masks equal the fp64 oracle's except at pre-activations within 1e-5 of the
     tensor's largest (rounding distance), and at most 1e-5 of all decisions differ;
  2. arithmetic: with the HIP path's decisions imposed on the oracle (`taps["force_masks"]`: same linear piece),
     forward output, EVERY weight / BatchNorm gradient, dx (worst entry, not a quantile) and the BatchNorm running
     statistics hold the flat 1e-3 of the tensor maximum (measured ~1e-6).
"""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import oracle.fill as OF  # noqa: E402
import oracle.visual as OV  # noqa: E402

TOL = 1e-3


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import textreid_amd  # noqa: F401

    return torch.device("cuda")


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def relmax(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


# (name, inplanes, planes, stride, H, W): every distinct Bottleneck shape of CLIP-RN50 at 384x128 (block_plan(RN50))
RN50_BLOCKS = [
    ("layer1.0", 64, 64, 1, 96, 32),
    ("layer1.1", 256, 64, 1, 96, 32),
    ("layer2.0", 256, 128, 2, 96, 32),
    ("layer2.1", 512, 128, 1, 48, 16),
    ("layer3.0", 512, 256, 2, 48, 16),
    ("layer3.1", 1024, 256, 1, 24, 8),
    ("layer4.0", 1024, 512, 1, 24, 8),
    ("layer4.1", 2048, 512, 1, 24, 8),
]


def _oracle_block(st, name, x, gout, stride, has_down, dtype, masks):
    s = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in st.items()}
    for k in s:
        if OV.is_param(k):
            s[k].requires_grad_(True)
    xr = x.to(dtype).clone().requires_grad_(True)
    taps = {"force_masks": list(masks)}
    out = OV.bottleneck(s, name, xr, stride, has_down, True, taps)
    out.backward(gout.to(dtype))
    return out.detach(), xr.grad, {k: v.grad for k, v in s.items() if OV.is_param(k)}, s, taps


FLIP_FRACTION = 1e-5  # at most this share of the ReLU decisions may differ from the fp64 oracle's ...
FLIP_MAGNITUDE = 1e-5  # ... and only at |pre-activation| <= this share of the tensor's largest


@pytest.mark.parametrize("path", ["p16", "split"])
@pytest.mark.parametrize("name,inpl,planes,stride,H,W", RN50_BLOCKS, ids=[b[0] for b in RN50_BLOCKS])
def test_bottleneck_b128_unstructured_masks(gpu, name, inpl, planes, stride, H, W, path):
    """path "p16": the production path - pre-split operands written by the BatchNorm passes (scales from the conv
    epilogue's column extremes / the backward bound), LDS-DMA staged GEMMs, transposing weight-gradient kernel;
    "split": the on-the-fly fp16 split (the path of encoders whose channel counts are not multiples of 32)."""
    from textreid_amd import ops
    from textreid_amd.backbones import m_resnet as M

    B, seed = 128, 11
    assert ops.conv_precision() == 16  # the bench's conv arithmetic
    blk = M.Bottleneck(inpl, planes, stride)
    has_down = blk.downsample is not None
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    # oracle names follow the reference's state dict: downsample.{0,1} for the conv / BatchNorm (the module's "-1" pool has no state)
    st = {}
    for k, shp in shapes.items():
        st[name + "." + k] = torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(name + "." + k, shp, seed)
    blk.load_state_dict({k: st[name + "." + k].clone() for k in shapes})
    blk = blk.to(gpu).train()
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    # block input = a post-ReLU activation (half zeros, unstructured); upstream gradient dense
    x = F.relu(_randn((B, inpl, H, W), seed))
    Ho, Wo = H // stride, W // stride
    gout = _randn((B, planes * 4, Ho, Wo), seed + 1)

    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    gd = gout.permute(0, 2, 3, 1).contiguous().to(gpu)
    WA = M.weight_amax(blk)
    ar = M.ConvArith(gpu, WA)
    ax = ops.amax(xd)
    nbt, masks = [], []
    ws = M._WgradStream(gpu)
    G = {}
    if path == "p16":
        assert M.p16_eligible(type("One", (), {"blocks": lambda self: [blk]})())
        outp, rec = M.block_forward_p16(blk, ops.p16_pack(xd, ax), M.p16_weights(blk, WA, False), gpu, True, True, nbt, masks)
        out, a_out = outp.unpack(), outp.amax
        dx, _ = M.block_backward_p16(blk, rec, gd, M.p16_weights(blk, WA, True), ws, G)
    else:
        out, a_out, rec = M.block_forward(blk, xd, ax, ar, True, True, nbt, masks)
        dx = M.block_backward(blk, rec, gd, ar, ws, G)
    ws.join()
    torch.cuda.synchronize()
    assert len(nbt) == (4 if has_down else 3) and len(masks) == 3

    o64, dx64, g64, s64, taps = _oracle_block(st, name, x, gout, stride, has_down, torch.float64,
                                              [m_.permute(0, 3, 1, 2).cpu() for m_ in masks])

    errs = {"out": relmax(out.permute(0, 3, 1, 2), o64), "dx": relmax(dx.permute(0, 3, 1, 2), dx64)}
    named = dict(blk.named_parameters())
    for k, p in named.items():
        errs["grad:" + k] = relmax(G[id(p)].reshape(p.shape), g64[name + "." + k])
    sd = blk.state_dict()
    for k in sd:
        if k.endswith(("running_mean", "running_var")):
            errs["state:" + k] = relmax(sd[k], s64[name + "." + k])
    amax_err = relmax(a_out, o64.abs().max())
    flips, total, fmax = taps.get("flips", 0), taps["relu_elems"], taps.get("flip_max_rel", 0.0)
    worst = max(errs, key=errs.get)
    print("%s B=%d: %d of %d ReLU decisions differ from the fp64 oracle (largest |pre-activation| there %.1e of the tensor's); "
          "same decisions: worst of %d quantities %.1e (%s), dx %.1e; amax(out) %.1e" % (
              name, B, flips, total, fmax, len(errs), errs[worst], worst, errs["dx"], amax_err))
    assert flips <= FLIP_FRACTION * total and fmax <= FLIP_MAGNITUDE, (flips, total, fmax)
    bad = {k: v for k, v in errs.items() if not v <= TOL}
    assert not bad, bad
    # the scalar the NEXT block's GEMM scales by: the exact maximum on the split path, an upper bound within 2x on the
    # P16 path (bound of the BatchNorm branch + bound of the residual; a looser scale costs one bit of the low plane)
    a, true = float(a_out), float(o64.abs().max())
    assert (true * (1 - 1e-6) <= a <= 2.0 * true) if path == "p16" else amax_err <= 1e-4, (a, true)


@pytest.mark.parametrize("path", ["p16", "split", "bf16"])
def test_stem_b128_unstructured_masks(gpu, path):
    """The stem at the benchmarked size, BatchNorm + ReLU with random masks, 2x2 average pool - forward, all nine
    gradients, running statistics against the oracle (m_resnet.py:198-207).  path "p16": the production path
    (csrc/stem_conv.hip: conv1 straight from the NCHW image on the exact fp32 MFMA, conv2 / conv3 and their data
    gradients on the ring-of-rows kernel over P16 tensors written by the BatchNorm passes, weight gradients on the
    transposing P16 kernel); "split": the general path (im2col + GEMM, on-the-fly fp16 split on 256x32 / 128x64 tiles);
    "bf16": the stem as configs[3]'s bf16 mode runs it - the SAME fp32-class kernels on the stem's own two P16 filters (the
    residual blocks' filters are bf16 there), output handed over as a bf16 tensor: gradients and statistics as in "p16", the
    output within one bf16 rounding of the oracle's."""
    from textreid_amd import ops
    from textreid_amd.backbones import m_resnet as M

    B, seed = 128, 12
    spec = OV.RN50
    m = M.ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width)
    keys = [k for k in m.state_dict() if k.split(".")[0] in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3")]
    st = {k: (torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(k, m.state_dict()[k].shape, seed)) for k in keys}
    m.load_state_dict({k: v.clone() for k, v in st.items()}, strict=False)
    m = m.to(gpu).train()
    images = _randn((B, 3, spec.height, spec.in_width), seed)
    gout = _randn((B, spec.width, spec.height // 4, spec.in_width // 4), seed + 1)
    ar = M.ConvArith(gpu, M.weight_amax(m))
    nbt, masks = [], []
    ws = M._WgradStream(gpu)
    G = {}
    gd = gout.permute(0, 2, 3, 1).contiguous().to(gpu)
    if path == "p16":
        imd = images.to(gpu)
        assert M.stem_p16_ok(m, imd)
        xp, rec = M.stem_forward_p16(m, imd, M.p16_weights(m, ar.WA, False), gpu, nbt, masks)
        x = xp.unpack()
        assert float(xp.amax) >= float(x.abs().max()) * (1 - 1e-6)  # the bound the residual blocks scale by
        M.stem_backward_p16(m, rec, gd, M.p16_weights(m, ar.WA, True), ws, G)
    elif path == "bf16":
        imd = images.to(gpu)
        assert M.stem_p16_ok(m, imd)
        WPs = M.p16_weights(m, ar.WA, False, 1, stem_only=True)
        assert len(WPs) == 2
        xb, rec = M.stem_forward_p16(m, imd, WPs, gpu, nbt, masks, out_fmt=2)
        assert xb.fmt == 2 and xb.data.dtype == torch.bfloat16
        x = xb.data.float()  # (the fp32-class output rounded once to bf16: 2^-9 relative, checked below at that level)
        M.stem_backward_p16(m, rec, gd, M.p16_weights(m, ar.WA, True, 1, stem_only=True), ws, G)
    else:
        x, ax, rec = M.stem_forward(m, images.to(gpu), ar, True, nbt, masks)
        M.stem_backward(m, rec, gd, ar, ws, G)
    ws.join()
    torch.cuda.synchronize()

    def oracle(dtype):
        s = {k: (v.to(dtype).clone() if v.dtype.is_floating_point else v.clone()) for k, v in st.items()}
        for k in s:
            if OV.is_param(k):
                s[k].requires_grad_(True)
        taps = {"force_masks": [m_.permute(0, 3, 1, 2).cpu() for m_ in masks]}
        y = images.to(dtype)
        y = OV._relu(OV._bn(s, "bn1", F.conv2d(y, s["conv1.weight"], stride=2, padding=1), True), taps)
        y = OV._relu(OV._bn(s, "bn2", F.conv2d(y, s["conv2.weight"], padding=1), True), taps)
        y = OV._relu(OV._bn(s, "bn3", F.conv2d(y, s["conv3.weight"], padding=1), True), taps)
        y = F.avg_pool2d(y, 2)
        y.backward(gout.to(dtype))
        return y.detach(), s, taps

    o64, s64, taps = oracle(torch.float64)
    errs = {"out": relmax(x.permute(0, 3, 1, 2), o64)}
    named = dict(m.named_parameters())
    for k in keys:
        if OV.is_param(k):
            errs["grad:" + k] = relmax(G[id(named[k])].reshape(named[k].shape), s64[k].grad)
        elif k.endswith(("running_mean", "running_var")):
            errs["state:" + k] = relmax(m.state_dict()[k], s64[k])
    flips, total, fmax = taps.get("flips", 0), taps["relu_elems"], taps.get("flip_max_rel", 0.0)
    print("stem (%s) B=%d: %d of %d ReLU decisions differ (|pre-activation| <= %.1e of max);" % (path, B, flips, total, fmax), {k: "%.1e" % v for k, v in errs.items()})
    assert flips <= FLIP_FRACTION * total and fmax <= FLIP_MAGNITUDE, (flips, total, fmax)
    bad = {k: v for k, v in errs.items() if not v <= (2.0 ** -8 if (path == "bf16" and k == "out") else TOL)}
    assert not bad, bad


BF16_BLOCKS = [RN50_BLOCKS[1], RN50_BLOCKS[2], RN50_BLOCKS[5]]  # identity / strided + downsample / deep identity


@pytest.mark.parametrize("name,inpl,planes,stride,H,W", BF16_BLOCKS, ids=[b[0] for b in BF16_BLOCKS])
def test_bottleneck_bf16_mode(gpu, name, inpl, planes, stride, H, W):
    """configs[3]'s bf16 mode on one Bottleneck (B = 32, He-style weights, unstructured masks): bf16 operands written
    by their producers, bf16 conv outputs, bf16 data gradients - against `oracle.visual.bf16_conv` (the definition of
    the mode) in fp64 with the HIP path's ReLU decisions imposed.  Rounding to bf16 is a step function, so two correct
    evaluations differ wherever an fp32-level difference straddles a rounding boundary; the bound is therefore the
    oracle's own fp32-vs-fp64 spread in that arithmetic (3x, floor 1e-3), as in test_config3_rn101_k65536_bf16."""
    from textreid_amd import ops
    from textreid_amd.backbones import m_resnet as M

    B, seed = 32, 13
    blk = M.Bottleneck(inpl, planes, stride)
    has_down = blk.downsample is not None
    shapes = {k: tuple(v.shape) for k, v in blk.state_dict().items()}
    st = {}
    for k, shp in shapes.items():
        st[name + "." + k] = torch.zeros((), dtype=torch.int64) if k.endswith("num_batches_tracked") else OF.fill(name + "." + k, shp, seed)
    blk.load_state_dict({k: st[name + "." + k].clone() for k in shapes})
    blk = blk.to(gpu).train()
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    bf = lambda t: t.to(torch.bfloat16).to(t.dtype)
    x = bf(F.relu(_randn((B, inpl, H, W), seed)))  # a block input IS a bf16 tensor in this mode
    Ho, Wo = H // stride, W // stride
    gout = _randn((B, planes * 4, Ho, Wo), seed + 1)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    gd = gout.permute(0, 2, 3, 1).contiguous().to(gpu).to(torch.bfloat16)  # the gradient of a bf16 block output
    WA = M.weight_amax(blk)
    nbt, masks, G = [], [], {}
    ws = M._WgradStream(gpu)
    outp, rec = M.block_forward_p16(blk, ops.p16_pack(xd, None, 2), M.p16_weights(blk, WA, False, 2), gpu, True, True, nbt, masks)
    assert rec[1].dtype == torch.bfloat16  # the conv outputs are stored as bf16 tensors
    dx, _ = M.block_backward_p16(blk, rec, gd, M.p16_weights(blk, WA, True, 2), ws, G)
    assert dx.dtype == torch.bfloat16
    ws.join()
    torch.cuda.synchronize()
    forced = [m_.permute(0, 3, 1, 2).cpu() for m_ in masks]

    def run(dtype):
        with OV.bf16_conv():
            o, dxr, g, s, taps = _oracle_block(st, name, x, gout, stride, has_down, dtype, forced)
        return o, bf(dxr), g, s, taps  # (a block input's gradient is rounded where that tensor was produced)

    o64, dx64, g64, s64, taps = run(torch.float64)
    o32, dx32, g32, s32, _ = run(torch.float32)
    named = dict(blk.named_parameters())
    sd = blk.state_dict()
    hip, spread = {"out": relmax(outp.unpack().permute(0, 3, 1, 2), o64), "dx": relmax(dx.float().permute(0, 3, 1, 2), dx64)}, \
                  {"out": relmax(o32, o64), "dx": relmax(dx32, dx64)}
    for k, p in named.items():
        hip["grad:" + k], spread["grad:" + k] = relmax(G[id(p)].reshape(p.shape), g64[name + "." + k]), relmax(g32[name + "." + k], g64[name + "." + k])
    for k in sd:
        if k.endswith(("running_mean", "running_var")):
            hip["state:" + k], spread["state:" + k] = relmax(sd[k], s64[name + "." + k]), relmax(s32[name + "." + k], s64[name + "." + k])
    ratio = {k: hip[k] / max(spread[k], TOL / 3) for k in hip}
    worst = sorted(ratio, key=ratio.get)[-3:]
    print("%s bf16 mode B=%d: %d ReLU decisions of %d differ from the fp64 oracle's; HIP error / oracle fp32-vs-fp64 spread: worst %s" % (
        name, B, taps.get("flips", 0), taps["relu_elems"], [(k, "%.1e vs %.1e" % (hip[k], spread[k])) for k in worst]))
    bad = {k: (hip[k], spread[k]) for k in hip if not hip[k] <= max(TOL, 3.0 * spread[k])}
    assert not bad, bad
