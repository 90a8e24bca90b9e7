"""Checkpoint ingestion: reference ``best.pth`` files and CLIP TorchScript archives.

``load_reference_state`` reproduces the loading rule of the reference's ``Checkpointer``
(``lib/utils/checkpoint.py:90-148``): a ``module.`` prefix left by DataParallel /
DistributedDataParallel is stripped when every key carries it, then every model key takes the
loaded entry whose name is its LONGEST SUFFIX (so ``embed_model.v_encoder_q.conv1.weight`` also
accepts a bare ``conv1.weight``), unmatched model keys keep their current value, and the result
is loaded strictly.  Host-side plumbing: runs once, no kernels.

``load_clip_visual`` is the TorchScript path of ``m_resnet.py:246-267``: ``torch.jit.load`` of
``RN50.pt`` / ``RN101.pt``, ``visual.`` prefix stripped and the 7x7 positional grid resized
(``backbones.m_resnet.state_filter``), non-strict load.
"""

import logging
from collections import OrderedDict

import torch


def strip_prefix_if_present(state_dict, prefix="module."):
    if not state_dict or not all(k.startswith(prefix) for k in state_dict):
        return state_dict
    return OrderedDict((k.replace(prefix, ""), v) for k, v in state_dict.items())


def match_by_longest_suffix(model_keys, loaded_keys):
    """{model key: loaded key} - the loaded key that is the longest suffix of the model key (checkpoint.py:90-103;
    ties between equally long suffixes cannot occur: equal-length suffixes of one string are the same string)."""
    by_len = sorted(loaded_keys, key=len, reverse=True)
    out = {}
    for k in model_keys:
        for cand in by_len:
            if k.endswith(cand):
                out[k] = cand
                break
    return out


def load_reference_state(model, loaded_state_dict, except_keys=None):
    """Load a reference-format state dict (or the ``model`` entry of a ``best.pth``) into ``model``."""
    logger = logging.getLogger("PersonSearch.checkpoint")
    if "model" in loaded_state_dict and not torch.is_tensor(loaded_state_dict["model"]):
        loaded_state_dict = loaded_state_dict["model"]
    loaded = strip_prefix_if_present(loaded_state_dict)
    state = model.state_dict()
    for key, src in match_by_longest_suffix(sorted(state), sorted(loaded)).items():
        if except_keys and any(e in key for e in except_keys):
            continue
        state[key] = loaded[src]
        logger.debug("%s loaded from %s of shape %s", key, src, tuple(loaded[src].shape))
    model.load_state_dict(state)
    return model


def load_checkpoint_file(model, path, except_keys=None, map_location="cpu"):
    """``Checkpointer.load`` counterpart (checkpoint.py:47-59): returns the checkpoint's non-model entries."""
    ckpt = torch.load(path, map_location=map_location)
    load_reference_state(model, ckpt.pop("model") if "model" in ckpt else ckpt, except_keys)
    return ckpt


def load_clip_visual(model, pretrained_path):
    """TorchScript CLIP archive -> image encoder (m_resnet.py:246-267)."""
    from .backbones.m_resnet import state_filter

    sd = torch.jit.load(pretrained_path, map_location="cpu").state_dict()
    missing, unexpected = model.load_state_dict(state_filter(sd, model.attnpool.spacial_dim), strict=False)
    return missing, unexpected
