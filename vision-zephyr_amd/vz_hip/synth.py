"""Deterministic synthetic weights and inputs for the Vision-Zephyr hot path.

There is no network and no checkpoint on either box, so every test, the smoke run
and the bench use weights generated from an integer hash of (seed, tensor name,
element index).  The generator uses only wrapping int32 torch arithmetic followed by ONE
fp32 multiply, so CPU and GPU produce bit-identical tensors and no weight file
ever has to travel.

Tensor names are the reference's own state-dict keys
(ref:vis_zephyr/model/language_model/vis_zephyr.py:37-45 builds `model.*` +
`lm_head`; Q-Former keys ref:vis_zephyr/model/multimodal_projector/builder.py:12-70;
CLIP keys are HF `CLIPVisionModel`'s under `model.vision_tower.vision_tower.`),
so the same dict loads into the reference with `load_state_dict` (oracle pinning)
and into the engine through the real-checkpoint path.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Tuple

import torch

_M32 = 0xFFFFFFFF


def _s32(c: int) -> int:
    """32-bit pattern -> signed int32 python int."""
    c &= _M32
    return c - (1 << 32) if c >= (1 << 31) else c


def _hash32_(x: torch.Tensor) -> torch.Tensor:
    """lowbias32 integer mix, in place on an int32 tensor (wrapping multiply, logical shifts
    emulated with masks).  Pure integer arithmetic: identical on CPU and GPU."""
    x.bitwise_xor_((x >> 16).bitwise_and_(0xFFFF))
    x.mul_(_s32(0x7FEB352D))
    x.bitwise_xor_((x >> 15).bitwise_and_(0x1FFFF))
    x.mul_(_s32(0x846CA68B))
    x.bitwise_xor_((x >> 16).bitwise_and_(0xFFFF))
    return x


def name_key(name: str, seed: int) -> int:
    return (zlib.crc32(name.encode()) * 0x9E3779B1 + seed * 0x85EBCA6B + 0x1234567) & _M32


def _hash_stream(key: int, start: int, stop: int, device) -> torch.Tensor:
    k = _s32(key)
    x = torch.arange(start, stop, dtype=torch.int32, device=device)
    x.bitwise_xor_(k)
    _hash32_(x)
    x.add_(k)
    _hash32_(x)
    return x


def hash_normal(name: str, shape, std: float, seed: int, device="cpu", mean: float = 0.0,
                chunk: int = 1 << 22) -> torch.Tensor:
    """fp32 tensor of `shape`, approximately N(mean, std^2) (Irwin-Hall of the hash's 4 bytes,
    bounded at +-3.45 sigma), bit-identical on every device."""
    n = 1
    for s in shape:
        n *= int(s)
    assert n < (1 << 31)
    out = torch.empty(n, dtype=torch.float32, device=device)
    key = name_key(name, seed)
    scale = float(std) / 147.8005413  # sqrt(4 * (256^2 - 1) / 12)
    for start in range(0, n, chunk):
        stop = min(n, start + chunk)
        x = _hash_stream(key, start, stop, device)
        s = x & 0xFF
        s.add_((x >> 8) & 0xFF)
        s.add_((x >> 16) & 0xFF)
        s.add_((x >> 24) & 0xFF)
        s.sub_(510)
        out[start:stop] = s.to(torch.float32).mul_(scale)
    if mean != 0.0:
        out += mean
    return out.view(*shape)


def hash_ids(name: str, n: int, lo: int, hi: int, seed: int) -> torch.Tensor:
    """int64 token ids uniform in [lo, hi)."""
    h = _hash_stream(name_key(name, seed), 0, n, "cpu").to(torch.int64) & _M32
    return lo + (h % (hi - lo))


# ----------------------------------------------------------------------------------------------
# Architecture description (ref:checkpoints/vis-zephyr-7b-v1-pretrain/config.json:1-39 plus the
# hard-coded Q-Former/CLIP constants)
# ----------------------------------------------------------------------------------------------
@dataclass
class ArchConfig:
    # Zephyr / Mistral
    hidden: int = 4096
    inter: int = 14336
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 8
    head_dim: int = 128
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    sliding_window: int = 4096
    # CLIP ViT-L/14-336
    clip_hidden: int = 1024
    clip_inter: int = 4096
    clip_layers: int = 24
    clip_heads: int = 16
    clip_image: int = 336
    clip_patch: int = 14
    clip_eps: float = 1e-5
    # Q-Former (ref:vis_zephyr/model/multimodal_projector/builder.py:49-70)
    qf_queries: int = 32
    qf_blocks: int = 8
    qf_heads: int = 8
    qf_kv_dim: int = 5120
    qf_eps: float = 1e-5
    # fusion (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:63-64)
    fusion_groups: int = 4
    fusion_layers_per_group: int = 5
    # mm_vision_select_feature == 'cls_patch' (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:66-73): CLS stays in
    clip_keep_cls: bool = False

    @property
    def clip_tokens(self) -> int:
        return (self.clip_image // self.clip_patch) ** 2 + 1

    @property
    def clip_patches(self) -> int:
        return (self.clip_image // self.clip_patch) ** 2

    @property
    def vision_tokens(self) -> int:
        """tokens per tile that leave the tower: 576 ('patch') or 577 ('cls_patch')"""
        return self.clip_patches + (1 if self.clip_keep_cls else 0)

    @property
    def qf_ffn(self) -> int:
        return self.hidden * 2

    def small(self, **kw) -> "ArchConfig":
        import dataclasses
        return dataclasses.replace(self, **kw)


VT = "model.vision_tower.vision_tower.vision_model."
QF = "model.mm_projector."


def param_specs(cfg: ArchConfig) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    """(name, shape, std, mean) for every parameter of the reference model, in state-dict naming."""
    sp: List[Tuple[str, Tuple[int, ...], float, float]] = []
    H, C = cfg.hidden, cfg.clip_hidden
    # --- CLIP vision tower ---
    sp.append((VT + "embeddings.class_embedding", (C,), 0.5, 0.0))
    sp.append((VT + "embeddings.patch_embedding.weight", (C, 3, cfg.clip_patch, cfg.clip_patch), 0.03, 0.0))
    sp.append((VT + "embeddings.position_embedding.weight", (cfg.clip_tokens, C), 0.3, 0.0))
    sp.append((VT + "pre_layrnorm.weight", (C,), 0.05, 1.0))
    sp.append((VT + "pre_layrnorm.bias", (C,), 0.05, 0.0))
    for i in range(cfg.clip_layers):
        p = VT + f"encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            sp.append((p + ln + ".weight", (C,), 0.05, 1.0))
            sp.append((p + ln + ".bias", (C,), 0.05, 0.0))
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sp.append((p + f"self_attn.{nm}.weight", (C, C), 0.04 if nm in ("q_proj", "k_proj") else 0.02, 0.0))
            sp.append((p + f"self_attn.{nm}.bias", (C,), 0.02, 0.0))
        sp.append((p + "mlp.fc1.weight", (cfg.clip_inter, C), 0.02, 0.0))
        sp.append((p + "mlp.fc1.bias", (cfg.clip_inter,), 0.02, 0.0))
        sp.append((p + "mlp.fc2.weight", (C, cfg.clip_inter), 0.02, 0.0))
        sp.append((p + "mlp.fc2.bias", (C,), 0.02, 0.0))
    sp.append((VT + "post_layernorm.weight", (C,), 0.05, 1.0))
    sp.append((VT + "post_layernorm.bias", (C,), 0.05, 0.0))
    # --- Q-Former ---
    sp.append((QF + "learned_queries", (cfg.qf_queries, H), 1.0, 0.0))
    sp.append((QF + "pre_norm.weight", (cfg.qf_kv_dim,), 0.05, 1.0))
    sp.append((QF + "pre_norm.bias", (cfg.qf_kv_dim,), 0.05, 0.0))
    sp.append((QF + "norm.weight", (H,), 0.05, 1.0))
    sp.append((QF + "norm.bias", (H,), 0.05, 0.0))
    for i in range(cfg.qf_blocks):
        p = QF + f"blocks.{i}."
        for ln in ("norm1", "norm2", "norm3"):
            sp.append((p + ln + ".weight", (H,), 0.05, 1.0))
            sp.append((p + ln + ".bias", (H,), 0.05, 0.0))
        sp.append((p + "self_attn.in_proj_weight", (3 * H, H), 0.015, 0.0))
        sp.append((p + "self_attn.in_proj_bias", (3 * H,), 0.02, 0.0))
        sp.append((p + "self_attn.out_proj.weight", (H, H), 0.01, 0.0))
        sp.append((p + "self_attn.out_proj.bias", (H,), 0.02, 0.0))
        sp.append((p + "cross_attn.q_proj_weight", (H, H), 0.015, 0.0))
        sp.append((p + "cross_attn.k_proj_weight", (H, cfg.qf_kv_dim), 0.015, 0.0))
        sp.append((p + "cross_attn.v_proj_weight", (H, cfg.qf_kv_dim), 0.01, 0.0))
        sp.append((p + "cross_attn.in_proj_bias", (3 * H,), 0.02, 0.0))
        sp.append((p + "cross_attn.out_proj.weight", (H, H), 0.01, 0.0))
        sp.append((p + "cross_attn.out_proj.bias", (H,), 0.02, 0.0))
        sp.append((p + "ffn.0.weight", (cfg.qf_ffn, H), 0.01, 0.0))
        sp.append((p + "ffn.0.bias", (cfg.qf_ffn,), 0.02, 0.0))
        sp.append((p + "ffn.2.weight", (H, cfg.qf_ffn), 0.01, 0.0))
        sp.append((p + "ffn.2.bias", (H,), 0.02, 0.0))
    # --- Zephyr / Mistral ---
    kvd = cfg.n_kv_heads * cfg.head_dim
    qd = cfg.n_heads * cfg.head_dim
    sp.append(("model.embed_tokens.weight", (cfg.vocab, H), 1.0, 0.0))
    for i in range(cfg.n_layers):
        p = f"model.layers.{i}."
        sp.append((p + "input_layernorm.weight", (H,), 0.05, 1.0))
        sp.append((p + "post_attention_layernorm.weight", (H,), 0.05, 1.0))
        sp.append((p + "self_attn.q_proj.weight", (qd, H), 0.025, 0.0))
        sp.append((p + "self_attn.k_proj.weight", (kvd, H), 0.025, 0.0))
        sp.append((p + "self_attn.v_proj.weight", (kvd, H), 0.02, 0.0))
        sp.append((p + "self_attn.o_proj.weight", (H, qd), 0.01, 0.0))
        sp.append((p + "mlp.gate_proj.weight", (cfg.inter, H), 0.02, 0.0))
        sp.append((p + "mlp.up_proj.weight", (cfg.inter, H), 0.02, 0.0))
        sp.append((p + "mlp.down_proj.weight", (H, cfg.inter), 0.01, 0.0))
    sp.append(("model.norm.weight", (H,), 0.05, 1.0))
    sp.append(("lm_head.weight", (cfg.vocab, H), 0.02, 0.0))
    return sp


def iter_state_dict(cfg: ArchConfig, seed: int = 0, device="cpu", prefixes=None
                    ) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (name, fp32 tensor) one at a time (the full model is 9.2 B parameters)."""
    for name, shape, std, mean in param_specs(cfg):
        if prefixes is not None and not any(name.startswith(p) for p in prefixes):
            continue
        yield name, hash_normal(name, shape, std, seed, device=device, mean=mean)


def state_dict(cfg: ArchConfig, seed: int = 0, device="cpu", prefixes=None) -> Dict[str, torch.Tensor]:
    return dict(iter_state_dict(cfg, seed, device, prefixes))


def synth_tiles(n_tiles: int, seed: int = 1, size: int = 336) -> torch.Tensor:
    """[N,3,size,size] fp32 ~ N(0,1): CLIP-normalised pixel statistics."""
    return hash_normal(f"tiles{n_tiles}", (n_tiles, 3, size, size), 1.0, seed)


def synth_ids(n_ids: int, vocab: int, image_pos: int = 5, seed: int = 2) -> torch.Tensor:
    """[n_ids] int64 ids uniform in [3, vocab) with one IMAGE_TOKEN_INDEX (-200) at `image_pos`
    (negative `image_pos` = no image sentinel)."""
    ids = hash_ids(f"ids{n_ids}", n_ids, 3, vocab, seed)
    if image_pos >= 0:
        ids[image_pos] = -200
    return ids
