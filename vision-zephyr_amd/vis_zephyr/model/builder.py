"""`load_pretrained_model` (a14) with the reference's signature and return contract
(ref:vis_zephyr/model/builder.py:16-161): returns (tokenizer, model, image_processor, context_length).

The three load modes of the reference are kept (LoRA adapter + base, base + mm_projector.bin, consolidated directory).
Weights are streamed file by file straight into the engine's HBM layout (bf16; the reference forces fp16,
builder.py:45); hub ids (`model_path`, `model_base`, `mm_vision_tower`) resolve through the local HuggingFace cache (`local_files_only`) - there is no network on the box."""
from __future__ import annotations

import json
import os
import warnings

from vz_hip import weights as W

from ..constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
from .language_model.vis_zephyr import VisZephyrConfig, VisZephyrForCausalLM


def _load_config(path: str) -> VisZephyrConfig:
    with open(os.path.join(path, "config.json")) as f:
        d = json.load(f)
    d.pop("model_type", None)
    d.pop("architectures", None)
    d.pop("transformers_version", None)
    return VisZephyrConfig(**d)


def load_pretrained_model(model_path, model_base, model_name, load_8bit: bool = False, load_4bit: bool = False,
                          device_map="auto", device="cuda", **kwargs):
    # load_4bit: the reference hands the LLM linears to bitsandbytes as NF4 with double quantisation, fp16 compute (ref builder.py:35-43).
    # Here: the same NF4 fake-quantisation of the decoder-layer linears (64-element blocks; absmax kept in fp32 - the double
    # quantisation of the scales is not restated, vz_hip/quant.py) at load time, computed with by the bf16 engine.
    # load_8bit: the reference quantises the LLM linears to int8 through bitsandbytes (ref builder.py:33-34); the MI355X
    # engine's 8-bit form is W8A16 - OCP e4m3 weights with one power-of-two scale per row, streamed by the decode GEMV
    # (vz_hip/quant.py); activations and the prefill GEMMs stay bf16
    if "zephyr" not in model_name.lower():
        raise ValueError(f"Unsupported model name: {model_name}. Only Zephyr models are supported at the moment.")
    from transformers import AutoTokenizer
    lora = "lora" in model_name.lower()
    if lora and model_base is None:
        warnings.warn("There is `lora` in model name but no `model_base` is provided. If you are loading a LoRA model, "
                      "please provide the `model_base` argument.")
        lora = False
    # hub ids (script/run_cli.sh passes HuggingFaceH4/zephyr-7b-beta; the shipped config names openai/clip-vit-large-patch14-336)
    # resolve through the LOCAL HF cache before anything raises
    model_path = W.resolve_hub_path(model_path, "model_path")
    if model_base is not None:
        model_base = W.resolve_hub_path(model_base, "model_base")
    tokenizer = AutoTokenizer.from_pretrained(model_base if model_base is not None else model_path, use_fast=False)
    config = _load_config(model_path)

    mm_use_im_start_end = getattr(config, "mm_use_im_start_end", False)
    if getattr(config, "mm_use_im_patch_token", True):
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    if mm_use_im_start_end:
        tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    config.vocab_size = len(tokenizer)                 # the engine is built at the resized vocabulary (32001)

    clip_dir = W.resolve_hub_path(getattr(config, "mm_vision_tower", None), "mm_vision_tower")
    dev = "cuda:0" if device == "cuda" else device
    model = VisZephyrForCausalLM(config, device=dev, max_ctx=kwargs.pop("max_ctx", 4096), weight_fp8=bool(load_8bit), weight_nf4=bool(load_4bit) and not load_8bit)
    model.load_state_dict_stream(W.resize_vocab(W.iter_reference_checkpoint(model_path, model_base, clip_dir, lora=lora),
                                                len(tokenizer)))
    tower = model.get_vision_tower()
    if not tower.is_loaded:
        tower.load_model()
    context_length = getattr(model.config, "max_squence_length", 2048)
    return tokenizer, model, tower.image_processor, context_length
