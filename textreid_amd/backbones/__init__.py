"""Encoders of the match path: CLIP ModifiedResNet (m_resnet.py) for images and the
bidirectional GRU (gru.py) for captions.  Each is one autograd node over an explicit list of
HIP kernel launches; the factories below pick one from the config's MODEL.*_MODEL name."""
from .build import build_textual_model, build_visual_model
from .gru import GRU
from .m_resnet import ModifiedResNet

__all__ = ["GRU", "ModifiedResNet", "build_textual_model", "build_visual_model"]
