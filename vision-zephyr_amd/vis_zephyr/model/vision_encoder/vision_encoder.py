"""CLIP vision tower facade over the native engine.

Mirrors the attribute surface callers use on the reference's tower
(ref:vis_zephyr/model/vision_encoder/vision_encoder.py:13-151): `is_loaded`, `load_model()`,
`image_processor`, `num_patches`, `hidden_size`, `config`, `dtype`, `device`, and `__call__(images)`
-> fused multi-layer features [T,576,5*1024].  The CLIP encoder, the feature selection and the
fusion all run inside `vz_clip_fused_features`; there is no torch.nn module here.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch


class CLIPVisionTower:
    def __init__(self, vision_tower_path, args, owner=None, delay_load=False):
        self.is_loaded = False
        self.vision_tower_path = "openai/clip-vit-large-patch14-336" if vision_tower_path is None else vision_tower_path
        self.select_feature = getattr(args, "mm_vision_select_feature", "patch")
        raw = getattr(args, "mm_vision_select_layer", None)
        if isinstance(raw, str):
            try:    # parsed for error parity only; the fusion always takes the last 21 hidden states (Appendix A Q1)
                self.select_layers = [int(x.strip()) for x in raw.split(",")]
            except ValueError:
                raise ValueError("Invalid format for mm_vision_select_layer. Expected a comma-separated string of "
                                 f"integers, but got: {raw}")
        else:
            self.select_layers = [-2]
        if self.select_feature not in ("patch", "cls_patch"):      # ref :72-73 raises at the first forward; same exception type
            raise ValueError(f"Unknown feature selection strategy: {self.select_feature}")
        if owner is not None and (self.select_feature == "cls_patch") != bool(owner.arch.clip_keep_cls):
            raise ValueError("mm_vision_select_feature changed after the engine was built: the number of visual tokens per "
                             "tile (576 'patch' / 577 'cls_patch') is fixed at construction")
        self._owner = owner            # VisZephyrForCausalLM holding the engine
        self.image_processor = None
        if not delay_load:
            self.load_model()

    # the engine's architecture constants stand in for CLIPVisionConfig
    @property
    def config(self):
        a = self._owner.arch
        return SimpleNamespace(hidden_size=a.clip_hidden, intermediate_size=a.clip_inter, image_size=a.clip_image,
                               patch_size=a.clip_patch, num_hidden_layers=a.clip_layers,
                               num_attention_heads=a.clip_heads, layer_norm_eps=a.clip_eps)

    def load_model(self):
        """ref vision_encoder.py:44-51 loads the CLIP weights + CLIPImageProcessor here (`delay_load`).  Same contract: if the
        engine has not been given the tower's weights yet (a model built by `from_pretrained`, whose directory holds the
        language model only), they are streamed in from `vision_tower_path` - a directory or a hub id resolved through the
        LOCAL HF cache; the image processor is host-side preprocessing and is created on demand."""
        eng = self._owner.engine
        if "clip.0.qkv.w" not in eng.w:
            from vz_hip import weights as W
            for name, t in W.iter_clip(W.resolve_hub_path(self.vision_tower_path, "mm_vision_tower")):
                eng.add_weight(name, t)
        if self.image_processor is None:
            self.image_processor = _make_image_processor(self.vision_tower_path, self._owner.arch.clip_image)
        self.is_loaded = True

    @torch.no_grad()
    def forward(self, images):
        self._owner._ensure_ready()
        eng = self._owner.engine
        if isinstance(images, list):
            return [eng.clip_fused_features(im if im.ndim == 4 else im.unsqueeze(0)).to(im.dtype) for im in images]
        if images.ndim == 3:
            images = images.unsqueeze(0)
        return eng.clip_fused_features(images).to(images.dtype if images.is_floating_point() else torch.bfloat16)

    __call__ = forward

    @property
    def dummy_feature(self):
        return torch.zeros(1, self.hidden_size, device=self.device, dtype=self.dtype)

    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self._owner.device

    @property
    def hidden_size(self):
        return self._owner.arch.clip_hidden * 5

    @property
    def num_patches(self):
        return (self.config.image_size // self.config.patch_size) ** 2


def _make_image_processor(path, size):
    """CLIPImageProcessor with OpenAI CLIP statistics (host-side; ref vision_encoder.py:47)."""
    try:
        from transformers import CLIPImageProcessor
    except Exception:       # transformers absent: preprocessing is the caller's business
        return None
    import os
    try:
        from vz_hip import weights as W
        path = W.resolve_hub_path(path, "mm_vision_tower")
    except FileNotFoundError:
        pass
    if os.path.isdir(path) and os.path.exists(os.path.join(path, "preprocessor_config.json")):
        return CLIPImageProcessor.from_pretrained(path)
    return CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size}, resample=3,
                              image_mean=[0.48145466, 0.4578275, 0.40821073],
                              image_std=[0.26862954, 0.26130258, 0.27577711], do_convert_rgb=True)
