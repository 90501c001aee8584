"""Real-checkpoint ingestion: stream (reference key, tensor) pairs out of the files the
reference's loader reads (ref:vis_zephyr/model/builder.py:53-138):

  * the Zephyr backbone directory: `*.safetensors` shards (or `pytorch_model*.bin`), keys `model.*`, `lm_head.weight`;
  * the HF CLIP directory: keys `vision_model.*` -> `model.vision_tower.vision_tower.vision_model.*`;
  * `mm_projector.bin`: flat torch.save dict, keys `model.mm_projector.*`;
  * optionally a LoRA adapter (`adapter_model.safetensors|bin` + `adapter_config.json`) merged on the fly
    (W += B @ A * alpha / r, what peft's merge_and_unload computes) and `non_lora_trainables.bin`.

Tensors are yielded one at a time so that the 14.5 GB backbone never sits in host memory twice; `Engine.load_weights`
casts each one to its HBM layout as it arrives.
"""
from __future__ import annotations

import glob
import json
import os
from typing import Dict, Iterator, Optional, Tuple

import torch

VT_PREFIX = "model.vision_tower.vision_tower."


def resolve_hub_path(name_or_path: Optional[str], what: str = "model") -> str:
    """A local directory is returned as is.  A hub id (`HuggingFaceH4/zephyr-7b-beta`, `openai/clip-vit-large-patch14-336` -
    ref:script/run_cli.sh:2, ref:checkpoints/vis-zephyr-7b-v1-pretrain/config.json:23) is looked up in the local HF cache
    ONLY (`local_files_only=True`: there is no network on the box), first through huggingface_hub, then by walking
    `$HF_HUB_CACHE | $HF_HOME/hub | ~/.cache/huggingface/hub` / models--org--name / snapshots / <newest>."""
    if name_or_path is None:
        raise FileNotFoundError(f"{what}: no path given")
    p = os.path.expanduser(str(name_or_path))
    if os.path.isdir(p):
        return p
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(repo_id=str(name_or_path), local_files_only=True)
    except Exception:
        pass
    roots = [os.environ.get("HF_HUB_CACHE"), os.environ.get("HUGGINGFACE_HUB_CACHE"),
             os.path.join(os.environ["HF_HOME"], "hub") if os.environ.get("HF_HOME") else None,
             os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]
    folder = "models--" + str(name_or_path).strip("/").replace("/", "--")
    for root in filter(None, roots):
        snaps = sorted(glob.glob(os.path.join(root, folder, "snapshots", "*")), key=os.path.getmtime)
        if snaps:
            return snaps[-1]
    raise FileNotFoundError(f"{what} = {name_or_path!r} is neither a local directory nor a snapshot in the local HuggingFace cache "
                            "(no network access: place the files in a directory, or populate the cache, and pass that)")


def _iter_file(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    if path.endswith(".safetensors"):
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            for k in f.keys():
                yield k, f.get_tensor(k)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        for k, v in sd.items():
            yield k, v


def _weight_files(directory: str):
    st = sorted(glob.glob(os.path.join(directory, "*.safetensors")))
    st = [f for f in st if not os.path.basename(f).startswith("adapter_")]
    if st:
        return st
    return sorted(f for f in glob.glob(os.path.join(directory, "pytorch_model*.bin")))


def iter_backbone(model_dir: str) -> Iterator[Tuple[str, torch.Tensor]]:
    files = _weight_files(model_dir)
    if not files:
        raise FileNotFoundError(f"no *.safetensors / pytorch_model*.bin weight files in {model_dir}")
    for f in files:
        yield from _iter_file(f)


def iter_clip(clip_dir: str) -> Iterator[Tuple[str, torch.Tensor]]:
    for k, v in iter_backbone(clip_dir):
        if k.startswith("vision_model."):
            yield VT_PREFIX + k, v
        elif k.startswith("text_model.") or k.startswith("visual_projection") or k.startswith("text_projection") or k == "logit_scale":
            continue        # a full CLIPModel checkpoint: only the vision tower is on the path
        else:
            yield VT_PREFIX + "vision_model." + k, v          # transformers 5.x CLIPVisionModel: keys without the prefix


def iter_projector(path: str) -> Iterator[Tuple[str, torch.Tensor]]:
    for k, v in _iter_file(path):
        k = k[len("base_model."):] if k.startswith("base_model.") else k
        k = k[len("model."):] if k.startswith("model.model.") else k
        yield k, v


def normalize_keys(named) -> Iterator[Tuple[str, torch.Tensor]]:
    """keys as torch.save'd by the trainer (`base_model.` / doubled `model.model.` prefixes) -> the reference's state-dict keys."""
    for k, v in named:
        k = k[len("base_model."):] if k.startswith("base_model.") else k
        k = k[len("model."):] if k.startswith("model.model.") else k
        yield k, v


def load_lora(adapter_dir: str) -> Dict[str, torch.Tensor]:
    """{reference weight key: delta} for every LoRA-adapted linear (delta = B @ A * alpha / r)."""
    cfg = json.load(open(os.path.join(adapter_dir, "adapter_config.json")))
    scale = cfg["lora_alpha"] / cfg["r"]
    files = [f for f in (os.path.join(adapter_dir, "adapter_model.safetensors"), os.path.join(adapter_dir, "adapter_model.bin"))
             if os.path.exists(f)]
    if not files:
        raise FileNotFoundError(f"no adapter_model.* in {adapter_dir}")
    a, b = {}, {}
    for k, v in _iter_file(files[0]):
        base = k.replace("base_model.model.", "", 1)
        if ".lora_A." in base:
            a[base.split(".lora_A.")[0] + ".weight"] = v.float()
        elif ".lora_B." in base:
            b[base.split(".lora_B.")[0] + ".weight"] = v.float()
    return {k: (b[k] @ a[k]) * scale for k in a if k in b}


def iter_reference_checkpoint(model_path: str, model_base: Optional[str], clip_dir: str,
                              lora: bool = False) -> Iterator[Tuple[str, torch.Tensor]]:
    """everything `load_pretrained_model` loads, in the reference's three modes (builder.py:53-129)."""
    deltas = load_lora(model_path) if lora else {}
    backbone = model_base if model_base is not None else model_path
    for k, v in iter_backbone(backbone):
        if k in deltas:
            v = (v.float() + deltas[k]).to(v.dtype)
        yield k, v
    if model_base is not None:
        proj = os.path.join(model_path, "non_lora_trainables.bin" if lora else "mm_projector.bin")
        if os.path.exists(proj):
            yield from iter_projector(proj)
        elif not lora:
            raise FileNotFoundError(f"{proj} not found")
    yield from iter_clip(clip_dir)


def resize_vocab(named: Iterator[Tuple[str, torch.Tensor]], new_vocab: int) -> Iterator[Tuple[str, torch.Tensor]]:
    """grow embed_tokens / lm_head to `new_vocab` rows the way HF's `resize_token_embeddings` does: new rows = mean of
    the old ones (ref:vis_zephyr/model/builder.py:141-153 adds `<im_patch>` => 32001 rows)."""
    for k, v in named:
        if k in ("model.embed_tokens.weight", "lm_head.weight") and v.shape[0] < new_vocab:
            extra = v.float().mean(0, keepdim=True).to(v.dtype).expand(new_vocab - v.shape[0], -1)
            v = torch.cat([v, extra], 0)
        yield k, v
