"""Top-level model, reference ``lib/models/model.py:8-45`` (MoCo embed head only)."""
from torch import nn

from .backbones import build_textual_model, build_visual_model
from .embeddings.moco_head.head import build_moco_head


class Model(nn.Module):
    def __init__(self, cfg, vocab_dict=None):
        super().__init__()
        self.visual_model = build_visual_model(cfg)
        self.textual_model = build_textual_model(cfg, vocab_dict=vocab_dict)
        if cfg.MODEL.EMBEDDING.EMBED_HEAD != "moco":
            raise NotImplementedError("only EMBED_HEAD='moco' is on the accelerated path (SURVEY section 2 #8)")
        self.embed_model = build_moco_head(cfg, self.visual_model, self.textual_model)
        self.embed_type = "moco"

    def forward(self, images, captions):
        return self.embed_model(images, captions)


def build_model(cfg, vocab_dict=None):
    return Model(cfg, vocab_dict=vocab_dict)
