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


def fill(name, shape, seed=0, dtype=torch.float32):
    """Value distribution is chosen from the tensor name so that activations
    stay O(1) through ~100 layers (He-style for convs / linears, near-identity
    BatchNorm, small biases)."""
    rs = _rs(name, seed)
    shape = tuple(int(s) for s in shape)
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


def fill_state(state, seed=0, prefix=""):
    """Return {name: fill(prefix+name, tensor.shape)} for a state dict; integer
    tensors (ids, pointers, counters) are passed through unchanged."""
    out = {}
    for k, v in state.items():
        if v.dtype.is_floating_point:
            out[k] = fill(prefix + k, v.shape, seed)
        else:
            out[k] = v.clone()
    return out


def randn(name, shape, seed=0, scale=1.0):
    rs = _rs("randn:" + name, seed)
    return torch.from_numpy((rs.standard_normal(tuple(shape)) * scale).astype(np.float32))


def randint(name, lo, hi, shape, seed=0):
    rs = _rs("randint:" + name, seed)
    return torch.from_numpy(rs.randint(lo, hi, size=tuple(shape)).astype(np.int64))
