from .build import build_textual_model, build_visual_model  # noqa: F401
