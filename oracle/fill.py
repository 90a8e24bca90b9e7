"""Deterministic, platform-independent tensor fill used by tests and fixtures.

Weights are never stored in fixtures: both sides (the imported reference in
the build container, the HIP path on the GPU box) fill every tensor of a
state dict from ``fill(name, shape, seed)``, which depends only on the tensor
name, its shape and the seed (numpy ``RandomState`` bit streams are frozen).
"""

import zlib

import numpy as np
import torch


def _rs(name, seed):
    h = (zlib.crc32(name.encode("utf-8")) + 0x9E3779B1 * (seed + 1)) & 0x7FFFFFFF
    return np.random.RandomState(h)


RELU_MIN = 2e-4  # smallest |ReLU input| a margin-style fixture accepts (fp32 forward error there is ~1e-5)
MARGIN = 8.0  # ReLU decision margin of the "margin" style, in standard deviations of the BatchNorm output


def _bn_margin(name, shape, seed):
    """BatchNorm affine of the ``margin`` style, or None when ``name`` is not a BatchNorm2d weight / bias.

    A deep random-weight ReLU network is ill-conditioned for GRADIENTS in fp32: with ~1e7 activations
    per image a few pre-activations always sit within rounding distance of zero, two fp32 evaluations
    with different summation orders take different sides of the ReLU there, and one flipped mask moves
    a row of a weight gradient by ~1/sqrt(#pixels) of its magnitude (the reference's own fp32 result is
    1-8 % from an fp64 evaluation under the ``he`` style, measured).  This style removes the cause, not
    the test: every BatchNorm feeding a ReLU gets beta = +-MARGIN * gamma with a per-channel sign, so a
    pre-activation is MARGIN sigma away from the ReLU kink (half the channels mostly on, half mostly
    off, a ~3e-7 tail of elements on the other side; at 4 sigma a B=16 RN50 step still saw ~1 flip per evaluation), and the residual-branch BatchNorms (bn3 /
    downsample) use a small gamma with ONE sign pattern per residual layer, so the sum with the
    identity path keeps the margin.  Measured: every one of the 161 (RN50) / 314 (RN101) parameter
    gradients of the fp32 reference agrees with fp64 to <= 8e-5, which is what makes a flat 1e-3 gate
    meaningful at full size."""
    parts = name.split(".")
    leaf = parts[-1]
    if len(shape) != 1 or leaf not in ("weight", "bias") or len(parts) < 2:
        return None
    mod = parts[-2]
    if not (mod.startswith("bn") or mod == "1"):  # "1" = downsample.1
        return None
    n = int(shape[0])
    bn = ".".join(parts[:-1])
    layer = [i for i, q in enumerate(parts) if q.startswith("layer") and q[5:].isdigit()]
    resid = bool(layer) and mod in ("bn3", "1")
    g = _rs(bn + ".gamma", seed).uniform(0.2, 0.4, size=n) if resid else _rs(bn + ".gamma", seed).uniform(0.8, 1.2, size=n)
    if leaf == "weight":
        return g
    tag = ".".join(parts[: layer[0] + 1]) if resid else bn
    sign = np.where(_rs("sign:" + tag, seed).uniform(size=n) < 0.5, -1.0, 1.0)
    return MARGIN * g * sign


def fill(name, shape, seed=0, dtype=torch.float32, style="he"):
    """Value distribution is chosen from the tensor name so that activations
    stay O(1) through ~100 layers (He-style for convs / linears, near-identity
    BatchNorm, small biases).  ``style="margin"`` replaces the BatchNorm affine
    parameters by the large-ReLU-margin choice of ``_bn_margin`` (full-size
    gradient fixtures); everything else is drawn exactly as in the default style."""
    shape = tuple(int(s) for s in shape)
    if style == "margin":
        a = _bn_margin(name, shape, seed)
        if a is not None:
            return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).to(dtype)
    elif style != "he":
        raise ValueError("unknown fill style %r" % (style,))
    rs = _rs(name, seed)
    n = int(np.prod(shape)) if shape else 1
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        a = rs.uniform(0.5, 1.5, size=n)
    elif leaf == "running_mean":
        a = rs.standard_normal(n) * 0.1
    elif leaf == "positional_embedding":
        a = rs.standard_normal(n) / np.sqrt(shape[-1])
    elif leaf == "bias":
        a = rs.standard_normal(n) * 0.05
    elif leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
        a = rs.uniform(0.6, 1.4, size=n)
    elif leaf in ("weight",) or leaf.startswith("weight_"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = rs.standard_normal(n) * np.sqrt(2.0 / max(fan_in, 1))
        if leaf.startswith("weight_"):  # GRU matrices: keep gates unsaturated
            a = rs.uniform(-1.0, 1.0, size=n) / np.sqrt(shape[1])
    elif leaf == "projection":
        a = rs.uniform(-1.0, 1.0, size=n) * np.sqrt(6.0 / (shape[0] + shape[1]))
    else:
        a = rs.standard_normal(n)
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape)).to(dtype)


def fill_state(state, seed=0, prefix="", style="he"):
    """Return {name: fill(prefix+name, tensor.shape)} for a state dict; integer
    tensors (ids, pointers, counters) are passed through unchanged."""
    out = {}
    for k, v in state.items():
        if v.dtype.is_floating_point:
            out[k] = fill(prefix + k, v.shape, seed, style=style)
        else:
            out[k] = v.clone()
    return out


def randn(name, shape, seed=0, scale=1.0):
    rs = _rs("randn:" + name, seed)
    return torch.from_numpy((rs.standard_normal(tuple(shape)) * scale).astype(np.float32))


def randint(name, lo, hi, shape, seed=0):
    rs = _rs("randint:" + name, seed)
    return torch.from_numpy(rs.randint(lo, hi, size=tuple(shape)).astype(np.int64))


# --------------------------------------------------------------------------- digests
# Fixtures cannot hold every activation / gradient of a full-size encoder (38 M gradient values
# for RN50), so they hold a DIGEST of each tensor: mean, mean |.|, max |.| and 16 values at
# indices drawn from the tensor's name (logical row-major index: NCHW activations, OIHW filters).
DIGEST_SAMPLES = 16


def digest(name, t):
    t = torch.as_tensor(t).detach().double().cpu().contiguous().view(-1)
    idx = _rs("digest:" + name, 0).randint(0, t.numel(), size=DIGEST_SAMPLES)
    head = torch.stack([t.mean(), t.abs().mean(), t.abs().max()])
    return torch.cat([head, t[torch.from_numpy(idx)]]).numpy()


def digest_err(got, ref, floor=1e-30):
    """Largest deviation of a digest from the reference digest, relative to the reference tensor's
    own scale: means against mean |.|, samples and max |.| against max |.|.  ``floor`` bounds that
    scale from below (see ``grad_floor``)."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    am, mx = max(ref[1], floor), max(ref[2], floor)
    return float(max(abs(got[0] - ref[0]) / am, abs(got[1] - ref[1]) / am, np.abs(got[2:] - ref[2:]).max() / mx))


def grad_floor(digests):
    """Scale floor for gradient comparisons: 1e-3 of the largest gradient entry in the model.  Some
    gradients are analytically ZERO - a BatchNorm bias whose channel is entirely on (or off) only shifts
    the input of the next 1x1 conv, which the following train-mode BatchNorm removes; the k_proj bias
    (softmax shift invariance) - so both sides hold rounding noise there and "relative to the tensor's
    own maximum" is meaningless.  Such tensors are compared against this floor instead."""
    return 1e-3 * max(float(np.asarray(d)[2]) for d in digests)


def strided_sample(g):
    """<= ~16k values of a filter / matrix gradient: rows and columns of its [N, rest] view at fixed strides."""
    g2 = torch.as_tensor(g).reshape(g.shape[0], -1)
    rs = max(1, g2.shape[0] // 96)
    cs = max(1, g2.shape[1] // 160)
    return g2[::rs, ::cs].contiguous()
