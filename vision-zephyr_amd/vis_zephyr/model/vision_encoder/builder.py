"""ref:vis_zephyr/model/vision_encoder/builder.py:8-24."""
import os

from .vision_encoder import CLIPVisionTower


def build_vision_tower(vision_tower_cfg, **kwargs):
    path = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if path is not None and (os.path.exists(path) or path.startswith("openai") or path.startswith("laion")):
        return CLIPVisionTower(vision_tower_path=path, args=vision_tower_cfg, **kwargs)
    raise ValueError(f"Unknown vision tower path: {path}")
