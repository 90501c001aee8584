#!/usr/bin/env python
"""Role timeline inside the one-launch attention half of a batch-1 decode layer (decode_fused.hip): when the attention, QKV-row
and O-row workgroups start, how long they wait for their producers, when they finish.

    python tools/fused_stamps.py [ctx] [layers]
Stamps are s_memrealtime (100 MHz) of the LAST launch of a few eager decode steps (the last layer of the last step), printed in
microseconds relative to the first workgroup's start."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vz_hip import binding as B, synth  # noqa: E402
from vz_hip.engine import Engine  # noqa: E402

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
MODE = int(sys.argv[3]) if len(sys.argv) > 3 else 1       # 1 = QKV + attention + O in one launch, 2 = attention + O
cfg = synth.ArchConfig(n_layers=layers, clip_layers=20)
eng = Engine(cfg, max_ctx=ctx + 64, max_tiles=1, max_text=8)
eng.load_synthetic(0)
ids = synth.synth_ids(ctx, cfg.vocab, image_pos=-1, seed=5)
x = eng.embed_tokens(ids).unsqueeze(0)
_, last = eng.prefill(x, [ctx])
eng.decode_begin(last.argmax(-1).to(torch.int32), [ctx], [ctx])
B.check(B.lib().vz_tune_set(12, MODE))                    # the one-launch form is opt-in
eng.decode_steps(32)                                   # graph replays: clocks / caches in their steady state
torch.cuda.synchronize()
B.check(B.lib().vz_tune_set(12, MODE))
B.check(B.lib().vz_tune_set(13, 1))
os.environ["VZ_NO_GRAPH"] = "1"                        # stamped launches run eagerly (the stamp buffer is a launch argument)
eng.decode_steps(4)
torch.cuda.synchronize()
B.check(B.lib().vz_tune_set(13, 0))
buf = (C.c_longlong * (4096 * 4))()
n = C.c_int(0)
B.check(B.lib().vz_prof_attn_half_stamps(buf, 4096, C.byref(n)))
st = np.ctypeslib.as_array(buf).reshape(4096, 4)[: n.value].astype(np.int64)
nA, nC = (6144 // 16 if MODE == 1 else 0), 4096 // 16
nB = n.value - nA - nC
t0 = st[:, 0][st[:, 0] > 0].min()
us = lambda a: (a - t0) / 100.0                        # noqa: E731
print(f"ctx {ctx}: {n.value} workgroups = {nB} attention + {nA} QKV rows + {nC} O rows; launch span {us(st.max()):.1f} us")
for name, lo, hi in (("attention", 0, nB), ("QKV rows", nB, nB + nA), ("O rows", nB + nA, n.value)):
    s = st[lo:hi]
    if hi <= lo:
        continue
    act = s[:, 0] > 0
    fin = s[:, 2] > 0
    line = f"  {name:10s}: start {us(s[act, 0]).min():6.1f} .. {us(s[act, 0]).max():6.1f}"
    w = s[:, 1] > 0
    if w.any():
        line += f" | wait over {us(s[w, 1]).min():6.1f} .. {us(s[w, 1]).max():6.1f}"
    if fin.any():
        line += f" | done {us(s[fin, 2]).min():6.1f} .. {us(s[fin, 2]).max():6.1f}  ({int(fin.sum())} workgroups)"
    early = s[:, 3] > 0
    if early.any():
        line += f" | non-merging splits left at {us(s[early, 3]).min():6.1f} .. {us(s[early, 3]).max():6.1f}"
    print(line)
