"""Loss evaluator of the MoCo head (reference ``moco_head/loss.py:8-43``).

Holds the one trainable tensor of the head that is not in an encoder - the [256, 11003]
identity ``projection`` (state-dict name ``loss_evaluator.projection``) - and evaluates the
three training losses on it.  ``forward`` takes the reference's materialised-logit arguments;
the head itself calls ``forward_fused``, which reads the queues in place.
"""
import torch
from torch import nn

from ... import losses as L

TEMPERATURE = 0.07  # loss.py:17, fixed in the reference (not a config key)


class LossComputation(nn.Module):
    T = TEMPERATURE

    def __init__(self, cfg):
        super().__init__()
        emb = cfg.MODEL.EMBEDDING
        # randn first, then xavier: draws from the global RNG in the reference's order (loss.py:11-18),
        # so a seeded run initialises every later module identically
        w = torch.randn(emb.FEATURE_SIZE, cfg.MODEL.NUM_CLASSES)
        nn.init.xavier_uniform_(w, gain=1)
        self.projection = nn.Parameter(w)
        self.epsilon = emb.EPSILON

    def _shared(self, v_embed, t_embed, labels):
        inst = L.instance_loss(self.projection, v_embed, t_embed, labels, epsilon=self.epsilon)
        return inst, L.global_align_loss(v_embed, t_embed, labels)

    @staticmethod
    def _pack(inst, nce, align):
        return {"instance_loss": inst, "infonce_loss": nce, "global_align_loss": align}

    def forward(self, v_embed, t_embed, v_pos, v_neg, t_pos, t_neg, labels):
        inst, align = self._shared(v_embed, t_embed, labels)
        return self._pack(inst, L.infonce_loss(v_pos, v_neg, t_pos, t_neg, self.T), align)

    def forward_fused(self, v_embed, t_embed, v_q, t_q, v_k, t_k, labels, t_queue, v_queue, id_queue):
        """Queue logits, the same-id mask and the cross entropy in one pass over the queues
        (no [B, |neg|] gather as head.py:101-115 builds)."""
        inst, align = self._shared(v_embed, t_embed, labels)
        nce = L.queue_infonce_loss(v_q, t_q, v_k, t_k, labels, t_queue, v_queue, id_queue, self.T)
        return self._pack(inst, nce, align)


def make_loss_evaluator(cfg):
    return LossComputation(cfg)
