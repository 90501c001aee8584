#!/usr/bin/env python
"""Decode rate of the bench request (BASELINE configs[2]: 5 tiles + 1889 ids -> S = 2048, 128 greedy tokens, 32 layers) under a
list of vz_tune_set settings, ONE engine for all of them (the graph re-captures when a frozen knob changes).
    python tools/bench_decode.py "" "17=0" "17=24,18=40" ...      (VZ_FP8=1: the W8A16 engine; VZ_LAYERS, VZ_REPS)
Each setting also prints the per-class launch times of 8 eager steps (GEMV / decode attention)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from vz_hip import binding as B, synth  # noqa: E402

layers = int(os.environ.get("VZ_LAYERS", "32"))
reps = int(os.environ.get("VZ_REPS", "2"))
fp8 = os.environ.get("VZ_FP8", "0") == "1"
n_tiles, n_ids, n_new = 5, 1889, 128
S = (n_ids - 1) + 32 * n_tiles
model = build_model(layers, "cuda:0", max_ctx=S + n_new + 16, weight_fp8=fp8)
eng, cfg = model.engine, model.arch
tiles = synth.synth_tiles(n_tiles, seed=1).to("cuda:0", torch.bfloat16)
ids = synth.synth_ids(n_ids, cfg.vocab, image_pos=5, seed=2).unsqueeze(0).to("cuda:0")
emb = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles])[4]
ref_ids = None
applied = {}
for setting in (sys.argv[1:] or [""]):
    for kv in filter(None, setting.split(",")):
        k, v = (int(t) for t in kv.split("="))
        B.check(B.lib().vz_tune_set(k, v))
        applied[k] = v

    def one():
        tm = {}
        t0 = time.perf_counter()
        out = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, eos_token_id=None, pad_token_id=2,
                             use_cache=True, timing=tm)
        torch.cuda.synchronize()
        return out, tm["t_first_token"] - t0, time.perf_counter() - tm["t_first_token"]
    out, _, _ = one()
    r = [one()[1:] for _ in range(reps)]
    same = "first" if ref_ids is None else ("ids equal" if torch.equal(out, ref_ids) else "IDS DIFFER")
    ref_ids = out if ref_ids is None else ref_ids
    tok = (n_new - 1) * reps / sum(b for _, b in r)
    line = f"[{setting or 'default'}] {tok:7.1f} tok/s  {1e3 / tok:6.3f} ms/token  first-token {sum(a for a, _ in r) / reps * 1e3:6.2f} ms  ({same})"
    _, last = eng.prefill(emb, [S])
    for klass, name in ((B.K_GEMV, "gemv"), (B.K_ATTN_DEC, "attn")):
        eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
        eng.prof_enable(True, klass)
        eng.decode_steps(8)
        torch.cuda.synchronize()
        n_l, ms = eng.prof_read()
        eng.prof_enable(False)
        line += f"  {name} {ms / max(1, n_l) * 1e3:6.2f} us x {n_l // 8}"
    print(line, flush=True)
