#!/usr/bin/env python
"""SURVEY config 5 in miniature (one GPU): VCR-style items - a 1920 x 804 frame (anyres -> 4 tiles) + a ~200-token prompt, 128 new
tokens - through device preprocessing -> CLIP / fusion / Q-Former -> batched greedy generation.  Prints items/s and decode
tokens/s for batch sizes 1..64 (bf16 weights, or `fp8` for the W8A16 engine).

    python tools/bench_vcr.py [fp8 | fp8mfma] [layers]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vz_hip import synth  # noqa: E402
from vz_hip.preprocess import AnyresPreprocessor  # noqa: E402

FP8 = "fp8" in sys.argv[1:] or "fp8mfma" in sys.argv[1:]
FP8_MFMA = "fp8mfma" in sys.argv[1:]        # W8A16 engine with its Zephyr prefill linears on the fp8 MFMA (Engine.set_prefill_fp8)
layers = next((int(a) for a in sys.argv[1:] if a.isdigit()), 32)
PINS = [[336, 672], [672, 336], [336, 1008], [1008, 336], [672, 672]]
N_NEW, L_PROMPT, MAXB = 128, 200, 64

from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM  # noqa: E402

hf = VisZephyrConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=8,
                     vocab_size=32000, rms_norm_eps=1e-5, sliding_window=4096, eos_token_id=2, pad_token_id=2, bos_token_id=1)
hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
hf.mm_patch_merge_type = "flat"
hf.image_aspect_ratio = "anyres"
hf.mm_grid_pinpoints = str(PINS)
hf.mm_hidden_size = 5120
model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, device="cuda:0", max_batch=MAXB, max_ctx=L_PROMPT + 4 * 32 + N_NEW + 16,
                                            max_tiles=4 * MAXB, max_text=L_PROMPT + 8, weight_fp8=FP8)
if FP8_MFMA:
    model.engine.set_prefill_fp8(True)
pre = AnyresPreprocessor("cuda:0")
rng = np.random.default_rng(0)
frames = [torch.from_numpy(rng.integers(0, 256, (804, 1920, 3), dtype=np.uint8)) for _ in range(MAXB)]     # host memory, as a loader hands them over
prompts = [synth.synth_ids(L_PROMPT, 32000, image_pos=5, seed=100 + i) for i in range(MAXB)]

for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):       # experiments: "knob=value,..." for vz_tune_set
    from vz_hip import binding as _B
    _B.check(_B.lib().vz_tune_set(*(int(v) for v in kv.split("="))))
ROWS = tuple(int(v) for v in os.environ.get("VZ_ROWS", "1,4,8,16,32,64").split(","))
for B in (() if "stream" in sys.argv[1:] else ROWS):
    def run():
        t0 = time.perf_counter()
        tiles = [pre(f, PINS) for f in frames[:B]]                          # H2D + LANCZOS + tiling + normalise on the device
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ids = torch.stack(prompts[:B]).to(model.device)
        tm = {}
        out = model.generate(input_ids=ids, images=tiles, images_size=[(1920, 804)] * B, do_sample=False, max_new_tokens=N_NEW,
                             eos_token_id=None, pad_token_id=2, timing=tm)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert tuple(out.shape) == (B, N_NEW) and all(t.shape[0] == 4 for t in tiles)
        return t1 - t0, tm["t_first_token"] - t1, t2 - tm["t_first_token"]
    run()
    best = min((run() for _ in range(2)), key=sum)
    tot = sum(best)
    print(f"{'fp8mfma ' if FP8_MFMA else 'fp8 ' if FP8 else ''}batch {B:2d}: {B / tot:6.2f} items/s  ({tot * 1e3:7.1f} ms per batch: preprocess {best[0] * 1e3:5.1f}, "
          f"image->first-token {best[1] * 1e3:6.1f}, decode {best[2] * 1e3:7.1f} = {B * (N_NEW - 1) / best[2]:7.1f} tok/s)", flush=True)

# ---- answers of different lengths (16..128 tokens, as eos would end them): static batches wait for their longest row,
# ---- continuous batching re-arms a row as soon as its sequence ends ----
if "stream" in sys.argv[1:]:
    ROWS = int(os.environ.get("VZ_STREAM_ROWS", "16"))        # rows decoding together; six batches' worth of items
    n_items = 6 * ROWS
    budgets = [int(b) for b in np.random.default_rng(1).integers(16, 129, n_items)]
    tiles_all = [pre(frames[i % MAXB], PINS) for i in range(n_items)]
    reqs = [{"input_ids": prompts[i % MAXB].unsqueeze(0), "images": tiles_all[i], "images_size": (1920, 804), "max_new_tokens": budgets[i]}
            for i in range(n_items)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_tok = 0
    for i0 in range(0, n_items, ROWS):        # static: every batch runs to its longest budget
        ids = torch.stack([r["input_ids"][0] for r in reqs[i0:i0 + ROWS]]).to(model.device)
        out = model.generate(input_ids=ids, images=[r["images"] for r in reqs[i0:i0 + ROWS]], do_sample=False,
                             max_new_tokens=max(budgets[i0:i0 + ROWS]), eos_token_id=None, pad_token_id=2)
        n_tok += sum(budgets[i0:i0 + ROWS])
    torch.cuda.synchronize()
    t_static = time.perf_counter() - t0
    res = {}
    for admit in (0, None):                   # one prefill per admission / the admissions of one sync share a prefill in the spare cache rows
        t0 = time.perf_counter()
        done = dict(model.generate_stream(reqs, eos_token_id=-1, rows=ROWS, sync_every=16, admit=admit))   # -1: no token ends a sequence (None = the config's eos)
        torch.cuda.synchronize()
        res[admit] = time.perf_counter() - t0
        assert len(done) == n_items and all(len(done[i]) == budgets[i] for i in range(n_items))
    print(f"{'fp8 ' if FP8 else ''}{n_items} items, budgets 16..128 tokens ({n_tok} useful tokens), {ROWS} rows: static batches {n_items / t_static:6.2f} items/s, "
          f"continuous batching {n_items / res[0]:6.2f} items/s ({t_static / res[0]:4.2f}x) with single admissions, "
          f"{n_items / res[None]:6.2f} items/s ({t_static / res[None]:4.2f}x) with batched admissions", flush=True)
