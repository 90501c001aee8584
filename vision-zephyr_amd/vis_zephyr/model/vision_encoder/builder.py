"""Factory for the vision tower handle (behaviour of ref:vis_zephyr/model/vision_encoder/builder.py:8-24).

A tower is named either by a local directory (config.json + preprocessor_config.json + weights) or by one of the two hub
families the reference accepts by prefix; anything else is rejected with the reference's ValueError.  The handle it returns
only carries configuration: the CLIP arithmetic itself runs inside the HIP engine (vz_clip_fused_features)."""
import os

from .vision_encoder import CLIPVisionTower

_HUB_PREFIXES = ("openai", "laion")


def _tower_name(cfg):
    # `mm_vision_tower` wins whenever the attribute exists (even if it is None), exactly as the reference's nested getattr
    if hasattr(cfg, "mm_vision_tower"):
        return cfg.mm_vision_tower
    return getattr(cfg, "vision_tower", None)


def build_vision_tower(vision_tower_cfg, **kwargs):
    name = _tower_name(vision_tower_cfg)
    known = name is not None and (os.path.exists(name) or name.startswith(_HUB_PREFIXES))
    if not known:
        raise ValueError(f"Unknown vision tower path: {name}")
    return CLIPVisionTower(vision_tower_path=name, args=vision_tower_cfg, **kwargs)
