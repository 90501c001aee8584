#!/usr/bin/env python
"""What e4m3 ACTIVATIONS cost the Zephyr prefill at full depth: one weight_fp8 engine (32 layers), the same embeddings through (a) its bf16 MFMA
prefill on the dequantised weights and (b) the fp8 MFMA prefill (`set_prefill_fp8`, per-row activation scales) - relative L2 of the logits, top-1
agreement per position, and the gap statistics where the top-1 differs.     python tools/fp8_prefill_depth.py [rows] [layers]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from vz_hip import binding as B, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = build_model(layers, "cuda:0", max_ctx=S + 16, weight_fp8=True)
eng, cfg = model.engine, model.arch
ids = synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=23).to("cuda:0")
emb = eng.embed_tokens(ids).unsqueeze(0)
a, _ = eng.prefill(emb, [S], all_logits=True, last_logits=False)
eng.set_prefill_fp8(True)
B.check(B.lib().vz_tune_set(22, 0))
b, _ = eng.prefill(emb, [S], all_logits=True, last_logits=False)
a, b = a[0].float(), b[0].float()
rel = float((a - b).norm() / a.norm())
rel_last = float((a[-1] - b[-1]).norm() / a[-1].norm())
same = (a.argmax(-1) == b.argmax(-1))
top2 = a.topk(2, -1).values
gap = (top2[:, 0] - top2[:, 1]) / a.pow(2).mean(-1).sqrt()
print(f"{layers} layers, {S} rows: fp8-MFMA prefill vs bf16 prefill of the same W8A16 engine: rel L2 all rows {rel:.4f}, last row {rel_last:.4f}; "
      f"top-1 equal on {int(same.sum())} / {S} rows; where it differs the bf16 path's own top-2 gap / rms is median "
      f"{float(gap[~same].median()) if (~same).any() else 0:.3f} (all rows: {float(gap.median()):.3f})")
