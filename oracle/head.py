"""Oracle: cross-modal MoCo head, one training step / eval encode.

Follows reference ``lib/models/embeddings/moco_head/head.py``:
  embed + normalise prologue  :111-130
  momentum (EMA) update       :73-94    (parameters only, BN buffers untouched)
  key forward                 :132-145  (train-mode BN, shared embed layers when FC=False)
  negative filter             :148-157  (one shared column set for the whole batch)
  pos / neg logits            :159-170
  enqueue                     :96-109
  eval path                   :178-183
and ``moco_head/loss.py:21-39`` (T=0.07, three losses).  ``FC=False`` only (the
shipped MoCo configs).  State is a flat dict keyed as the reference's
``embed_model.*`` state dict, queues in reference layout [C,K].
Test infrastructure only.
"""

import torch
import torch.nn.functional as F

from . import losses as L
from . import text as T
from . import visual as V

TEMPERATURE = 0.07  # moco_head/loss.py:18


def sub(state, prefix):
    """View of the entries under ``prefix.`` (shares tensors)."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in state.items() if k.startswith(prefix + ".")}


class _Sub(dict):
    """dict view that writes rebinding assignments back to the parent state."""

    def __init__(self, state, prefix):
        super().__init__(sub(state, prefix))
        self._state, self._prefix = state, prefix

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        self._state[self._prefix + "." + k] = v


def state_shapes(spec, K, C=256, num_classes=11003, hidden=512, embed=512, fc=False):
    sh = {}
    for enc in ("v_encoder_q", "v_encoder_k"):
        for k, s in V.state_shapes(spec).items():
            sh[enc + "." + k] = s
    for enc in ("t_encoder_q", "t_encoder_k"):
        for k, s in T.state_shapes(hidden, embed).items():
            sh[enc + "." + k] = s
    if fc:  # head.py:32-49 (module order of the reference: fc heads before the embed layers)
        for nm, cin in (("v_fc_q", spec.output_dim), ("t_fc_q", 2 * hidden), ("v_fc_k", spec.output_dim), ("t_fc_k", 2 * hidden)):
            sh[nm + ".0.weight"], sh[nm + ".0.bias"] = (C, cin), (C,)
            sh[nm + ".2.weight"], sh[nm + ".2.bias"] = (C, C), (C,)
    sh["v_embed_layer.weight"] = (C, spec.output_dim)
    sh["v_embed_layer.bias"] = (C,)
    sh["t_embed_layer.weight"] = (C, 2 * hidden)
    sh["t_embed_layer.bias"] = (C,)
    sh["t_queue"] = (C, K)
    sh["v_queue"] = (C, K)
    sh["id_queue"] = (1, K)
    sh["queue_ptr"] = (1,)
    sh["loss_evaluator.projection"] = (C, num_classes)
    return sh


def trainable_names(state):
    """Names that receive gradients (query encoders, embed layers, projection)."""
    out = []
    for k, v in state.items():
        if not v.dtype.is_floating_point:
            continue
        if k.startswith(("v_encoder_k.", "t_encoder_k.", "v_fc_k.", "t_fc_k.")) or k in ("t_queue", "v_queue"):
            continue
        if not V.is_param(k):
            continue
        out.append(k)
    return out


def ema_names(state):
    """(q_name, k_name) pairs touched by the momentum update, reference order."""
    pairs = []
    for enc in ("v_encoder", "t_encoder", "v_fc", "t_fc"):  # fc heads only exist with MODEL.MOCO.FC (head.py:86-94)
        for k in state:
            if k.startswith(enc + "_q.") and V.is_param(k):
                pairs.append((k, enc + "_k." + k[len(enc) + 3 :]))
    return pairs


@torch.no_grad()
def momentum_update(state, m):
    # head.py:78-85: param_k = param_k*m + param_q*(1-m)
    for qn, kn in ema_names(state):
        state[kn] = state[kn] * m + state[qn].detach() * (1.0 - m)


def encode(state, which, spec, table, images, tokens, lengths, training, vtaps=None):
    v = V.visual_forward(_Sub(state, "v_encoder_" + which), images, spec, training, vtaps)
    t = T.text_forward(sub(state, "t_encoder_" + which), table, tokens, lengths)
    return v, t


def fc_pair(state, which, v_feat, t_feat):
    """The MOCO.FC projection heads: Linear -> ReLU -> Linear (head.py:33-42)."""
    out = []
    for nm, x in (("v_fc_" + which, v_feat), ("t_fc_" + which, t_feat)):
        h = F.relu(F.linear(x, state[nm + ".0.weight"], state[nm + ".0.bias"]))
        out.append(F.linear(h, state[nm + ".2.weight"], state[nm + ".2.bias"]))
    return out


def embed_pair(state, v_feat, t_feat):
    v = F.linear(v_feat, state["v_embed_layer.weight"], state["v_embed_layer.bias"])
    t = F.linear(t_feat, state["t_embed_layer.weight"], state["t_embed_layer.bias"])
    return v, t


def negative_columns(id_queue, id_q):
    """head.py:148-157 restated: column k is a negative iff id_queue[k] equals
    no id in the batch.  Returns sorted int64 indices (what unique() yields)."""
    hit = (id_queue.view(1, -1) == id_q.view(-1, 1)).any(dim=0)
    return torch.nonzero(~hit, as_tuple=False).view(-1)


def contrast_logits(state, v_q, t_q, v_k, t_k, id_q):
    neg = negative_columns(state["id_queue"], id_q)
    v_pos = (v_q * t_k).sum(1, keepdim=True)
    v_neg = v_q @ state["t_queue"].detach()[:, neg]
    t_pos = (t_q * v_k).sum(1, keepdim=True)
    t_neg = t_q @ state["v_queue"].detach()[:, neg]
    return v_pos, v_neg, t_pos, t_neg


@torch.no_grad()
def enqueue(state, v_k, t_k, id_q):
    B = v_k.shape[0]
    K = state["v_queue"].shape[1]
    ptr = int(state["queue_ptr"])
    assert K % B == 0  # head.py:101
    state["v_queue"][:, ptr : ptr + B] = v_k.t()
    state["t_queue"][:, ptr : ptr + B] = t_k.t()
    state["id_queue"][:, ptr : ptr + B] = id_q.view(1, -1)
    state["queue_ptr"][0] = (ptr + B) % K


def losses_from_embeddings(state, v_embed, t_embed, v_q, t_q, v_k, t_k, id_q, epsilon):
    v_pos, v_neg, t_pos, t_neg = contrast_logits(state, v_q, t_q, v_k, t_k, id_q)
    return {
        "instance_loss": L.instance_loss(state["loss_evaluator.projection"], v_embed, t_embed, id_q, epsilon=epsilon),
        "infonce_loss": L.infonce_loss(v_pos, v_neg, t_pos, t_neg, TEMPERATURE),
        "global_align_loss": L.global_align_loss(v_embed, t_embed, id_q),
    }


def train_forward(state, spec, table, images, tokens, lengths, ids, m=0.999, epsilon=0.1, taps=None):
    """One MoCoHead.forward in training mode.  Mutates ``state`` exactly as the
    reference module mutates itself (BN running stats of all four encoders, key
    parameters, queues, pointer).  Returns the loss dict."""
    id_q = ids.long()
    vtaps = taps.setdefault("visual_q", {}) if taps is not None else None  # per-stage taps + ReLU margin of the query encoder
    v_feat, t_feat = encode(state, "q", spec, table, images, tokens, lengths, True, vtaps)
    fc = "v_fc_q.0.weight" in state
    v_embed, t_embed = embed_pair(state, v_feat, t_feat)
    vq_src, tq_src = fc_pair(state, "q", v_feat, t_feat) if fc else (v_embed, t_embed)  # head.py:117-129
    v_q, t_q = F.normalize(vq_src, dim=1), F.normalize(tq_src, dim=1)
    with torch.no_grad():
        momentum_update(state, m)
        vk_feat, tk_feat = encode(state, "k", spec, table, images, tokens, lengths, True)
        v_k, t_k = fc_pair(state, "k", vk_feat, tk_feat) if fc else embed_pair(state, vk_feat, tk_feat)
        v_k, t_k = F.normalize(v_k, dim=1), F.normalize(t_k, dim=1)
    out = losses_from_embeddings(state, v_embed, t_embed, v_q, t_q, v_k, t_k, id_q, epsilon)
    if taps is not None:
        taps.update(v_feat=v_feat, t_feat=t_feat, v_embed=v_embed, t_embed=t_embed, v_k=v_k, t_k=t_k)
    enqueue(state, v_k, t_k, id_q)
    return out


@torch.no_grad()
def eval_forward(state, spec, table, images, tokens, lengths):
    v_feat, t_feat = encode(state, "q", spec, table, images, tokens, lengths, False)
    return list(embed_pair(state, v_feat, t_feat))


def init_queues(state, seed=0):
    """rand -> normalize(dim=0), ids = -1, ptr = 0 (head.py:53-59), but drawn from
    the oracle's deterministic fill so both sides agree."""
    from .fill import _rs
    import numpy as np

    for nm in ("t_queue", "v_queue"):
        a = _rs("queue:" + nm, seed).uniform(0.0, 1.0, size=tuple(state[nm].shape)).astype(np.float32)
        state[nm] = F.normalize(torch.from_numpy(a), dim=0)
    state["id_queue"] = -torch.ones_like(state["id_queue"])
    state["queue_ptr"] = torch.zeros_like(state["queue_ptr"])
