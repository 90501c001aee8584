#!/usr/bin/env python
"""Stage cycle counts of one wave of the prefill attention kernel (vz_tune_set(16, 1) -> vz_prof_attn_stamps): where the longest causal
workgroup's wave 0 spends its shader cycles per 64-key tile.  python tools/attn_stamps.py [S]"""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd")); sys.path.insert(0, REPO)
import torch
from vz_hip import binding as B
B.load_library()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
Hq, Hkv, D = 32, 8, 128
q = torch.randn(1, S, Hq, D, device="cuda").bfloat16(); k = torch.randn(1, S, Hkv, D, device="cuda").bfloat16(); v = torch.randn(1, S, Hkv, D, device="cuda").bfloat16()
for _ in range(3):
    B.attention(q, k, v, D ** -0.5, True, 0, 4096)
torch.cuda.synchronize()
B.check(B.lib().vz_tune_set(16, 1))
B.attention(q, k, v, D ** -0.5, True, 0, 4096)
st = (C.c_longlong * 16)()
B.check(B.lib().vz_prof_attn_stamps(st))
B.check(B.lib().vz_tune_set(16, 0))
names = ["issue K/V global loads", "QK^T (2 halves: frag reads + 32 MFMA issue)", "softmax (incl. wait for the QK^T results)", "PV (tr-reads + 32 MFMA issue)",
         "wait loads + LDS store", "barrier", "-", "loop top"]
n = max(1, st[8]); tot = sum(st[i] for i in range(8))
print(f"S={S}: {n} tiles, {tot / n:.0f} cycles per tile for this wave (MFMA floor 1024 per wave, 2 waves per SIMD)")
for i, nm in enumerate(names):
    if st[i]: print(f"  {nm:48s} {st[i] / n:8.0f} cycles/tile  {100 * st[i] / tot:5.1f} %")
