from .builder import build_multimodal_projector, QFormer  # noqa: F401
