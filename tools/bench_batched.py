#!/usr/bin/env python
"""Batched greedy decode throughput: B text-only prompts of 64 ids, 64 new tokens each, Zephyr-7B shape.
    python tools/bench_batched.py [layers]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from vz_hip import synth  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
import bench  # noqa: E402


FP8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"


def build(max_batch):
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    hf = VisZephyrConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=32000, rms_norm_eps=1e-5, sliding_window=4096, eos_token_id=2, pad_token_id=2,
                         bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    return VisZephyrForCausalLM.from_synthetic(hf, seed=0, device="cuda:0", max_batch=max_batch, max_ctx=512, max_tiles=1, max_text=64,
                                               weight_fp8=FP8)


model = build(64)
for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):       # experiments: "knob=value,..." for vz_tune_set
    from vz_hip import binding as _B
    _B.check(_B.lib().vz_tune_set(*(int(v) for v in kv.split("="))))
ROWS = [int(v) for v in os.environ.get("VZ_ROWS", "1,2,4,8,16,32,48,64").split(",")]
n_new = 64
for B in ROWS:
    ids = torch.stack([synth.synth_ids(64, 32000, image_pos=-1, seed=10 + b) for b in range(B)]).cuda()
    model.generate(input_ids=ids, do_sample=False, max_new_tokens=n_new, eos_token_id=None, pad_token_id=2)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        tm = {}
        t0 = time.perf_counter()
        out = model.generate(input_ids=ids, do_sample=False, max_new_tokens=n_new, eos_token_id=None, pad_token_id=2, timing=tm)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        best = min(best, t1 - tm.get("t_first_token", t0))
    assert tuple(out.shape) == (B, n_new)
    print(f"{'fp8 ' if FP8 else ''}batch {B}: {B * (n_new - 1) / best:8.1f} decode tokens/s total ({(n_new - 1) / best:7.1f} per sequence), {best / (n_new - 1) * 1e3:6.3f} ms per step", flush=True)
