from .builder import build_vision_tower  # noqa: F401
from .vision_encoder import CLIPVisionTower  # noqa: F401
