"""Top-level module: two backbones and the MoCo embedding head that owns their query/key copies
(reference ``lib/models/model.py:8-45``; state-dict prefixes ``visual_model.``,
``textual_model.``, ``embed_model.`` are the reference's)."""
from torch import nn

from . import backbones
from .embeddings.moco_head.head import build_moco_head


class Model(nn.Module):
    embed_type = "moco"

    def __init__(self, cfg, vocab_dict=None):
        super().__init__()
        head = cfg.MODEL.EMBEDDING.EMBED_HEAD
        if head != self.embed_type:
            raise NotImplementedError(f"EMBED_HEAD={head!r}: only 'moco' is on the accelerated path (SURVEY 8)")
        self.visual_model = backbones.build_visual_model(cfg)
        self.textual_model = backbones.build_textual_model(cfg, vocab_dict=vocab_dict)
        self.embed_model = build_moco_head(cfg, self.visual_model, self.textual_model)

    def forward(self, images, captions):
        """Training: dict of losses.  Eval: (image embedding, caption embedding)."""
        return self.embed_model(images, captions)


def build_model(cfg, vocab_dict=None):
    return Model(cfg, vocab_dict=vocab_dict)
