"""Backbone dispatch, reference ``lib/models/backbones/build.py:6-17``."""
from .gru import build_gru
from .m_resnet import build_m_resnet


def build_visual_model(cfg):
    if cfg.MODEL.VISUAL_MODEL in ["m_resnet50", "m_resnet101", "m_resnet"]:
        return build_m_resnet(cfg)
    # torchvision-style resnet50/101 baselines are outside the accelerated path
    raise NotImplementedError(cfg.MODEL.VISUAL_MODEL)


def build_textual_model(cfg, vocab_dict=None):
    if cfg.MODEL.TEXTUAL_MODEL == "bigru":
        return build_gru(cfg, bidirectional=True, vocab_dict=vocab_dict)
    raise NotImplementedError(cfg.MODEL.TEXTUAL_MODEL)
