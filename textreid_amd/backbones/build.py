"""Backbone factories keyed by the config's model names (reference
``lib/models/backbones/build.py:6-17``).  Only the encoders on the accelerated path are
registered; any other name raises ``NotImplementedError`` as the reference does."""
from .gru import build_gru
from .m_resnet import build_m_resnet

_VISUAL = dict.fromkeys(("m_resnet", "m_resnet50", "m_resnet101"), build_m_resnet)
_TEXTUAL = {"bigru": lambda cfg, vocab_dict: build_gru(cfg, bidirectional=True, vocab_dict=vocab_dict)}


def _lookup(table, name):
    try:
        return table[name]
    except KeyError:
        # torchvision-style resnet50/101 and the BERT text encoder are outside the path
        raise NotImplementedError(name) from None


def build_visual_model(cfg):
    return _lookup(_VISUAL, cfg.MODEL.VISUAL_MODEL)(cfg)


def build_textual_model(cfg, vocab_dict=None):
    return _lookup(_TEXTUAL, cfg.MODEL.TEXTUAL_MODEL)(cfg, vocab_dict)
