"""Loss evaluator of the MoCo head, reference ``moco_head/loss.py:8-43``."""
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from ... import losses


class LossComputation(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.projection = Parameter(
            torch.randn(cfg.MODEL.EMBEDDING.FEATURE_SIZE, cfg.MODEL.NUM_CLASSES), requires_grad=True
        )
        self.epsilon = cfg.MODEL.EMBEDDING.EPSILON
        self.T = 0.07
        nn.init.xavier_uniform_(self.projection.data, gain=1)

    def forward(self, v_embed, t_embed, v_pos, v_neg, t_pos, t_neg, labels):
        """Reference signature (materialised logits)."""
        return {
            "instance_loss": losses.instance_loss(self.projection, v_embed, t_embed, labels, epsilon=self.epsilon),
            "infonce_loss": losses.infonce_loss(v_pos, v_neg, t_pos, t_neg, self.T),
            "global_align_loss": losses.global_align_loss(v_embed, t_embed, labels),
        }

    def forward_fused(self, v_embed, t_embed, v_q, t_q, v_k, t_k, labels, t_queue, v_queue, id_queue):
        """Same three losses with the queue logits fused (no [B,|neg|] gather)."""
        return {
            "instance_loss": losses.instance_loss(self.projection, v_embed, t_embed, labels, epsilon=self.epsilon),
            "infonce_loss": losses.queue_infonce_loss(v_q, t_q, v_k, t_k, labels, t_queue, v_queue, id_queue, self.T),
            "global_align_loss": losses.global_align_loss(v_embed, t_embed, labels),
        }


def make_loss_evaluator(cfg):
    return LossComputation(cfg)
