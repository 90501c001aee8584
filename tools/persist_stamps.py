#!/usr/bin/env python
"""Where a token of the persistent decode-token kernel (csrc/decode_persist.hip) spends its time: s_memrealtime stamps of workgroup 0's
sync wave at the 12 phase edges of every layer, averaged over the layers of the last decoded token of BASELINE configs[2]'s context.
    python tools/persist_stamps.py [layers]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from vz_hip import binding as B, synth  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
B.check(B.load_library().vz_tune_set(28, 1))            # opt in
for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):
    B.check(B.load_library().vz_tune_set(*(int(v) for v in kv.split("="))))
S = 2048
model = build_model(layers, "cuda:0", max_ctx=S + 144)
eng = model.engine
emb = eng.embed_tokens(synth.synth_ids(S, 32000, image_pos=-1, seed=3).unsqueeze(0))
_, last = eng.prefill(emb, [S])
eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
eng.decode_steps(32)
torch.cuda.synchronize()
eng.check_async()
assert eng.persist_mode(), "the persistent kernel did not run"
buf = (C.c_ulonglong * (layers * 12))()
B.check(eng.lib.vz_prof_persist_stamps(eng.h, buf, layers))
t = np.array(buf, dtype=np.float64).reshape(layers, 12) * 0.01        # microseconds
names = ["wait x (prev down)", "gather x + RMSNorm", "QKV units", "publish+wait+gather qkv", "attention (this workgroup)", "arrive+wait+gather att",
         "O units", "publish+wait+gather+norm x1", "gate|up units", "publish+wait+gather act", "down units"]
d = np.diff(t, axis=1)[1:]                      # layer 0 has no wait stamp
print(f"per layer, workgroup 0, mean over layers 1..{layers - 1} (us):")
for n, v in zip(names, d.mean(axis=0)):
    print(f"  {n:32s} {v:7.2f}")
per_layer = (t[1:, 11] - t[:-1, 11]).mean()
print(f"  layer period {per_layer:7.2f} us  -> {per_layer * layers / 1e3:.3f} ms per token in the layers")

