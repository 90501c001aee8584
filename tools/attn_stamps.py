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
st = (C.c_longlong * (16 + 8192))()
B.check(B.lib().vz_prof_attn_stamps(st))
B.check(B.lib().vz_tune_set(16, 0))
names = ["issue K/V global loads", "QK^T (2 halves: frag reads + 32 MFMA issue)", "softmax (incl. wait for the QK^T results)", "PV (tr-reads + 32 MFMA issue)",
         "wait loads + LDS store", "barrier", "-", "loop top"]
n = max(1, st[8]); tot = sum(st[i] for i in range(8))
print(f"S={S}: {n} tiles, {tot / n:.0f} cycles per tile for this wave (MFMA floor 1024 per wave, 2 waves per SIMD)")
for i, nm in enumerate(names):
    if st[i]: print(f"  {nm:48s} {st[i] / n:8.0f} cycles/tile  {100 * st[i] / tot:5.1f} %")

# ---- the schedule: which workgroups shared a CU, and when they ran ----
import collections
n_wg = (S + 127) // 128 * Hq
recs = [(st[16 + 4 * i], st[17 + 4 * i], st[18 + 4 * i], st[19 + 4 * i]) for i in range(n_wg) if st[17 + 4 * i]]
t0 = min(r[0] for r in recs); t1 = max(r[1] for r in recs)
print(f"{len(recs)} workgroups, kernel span {(t1 - t0) / 100:.1f} us (100 MHz stamps)")
by_cu = collections.defaultdict(list)
for i, (a, b, hw, qt) in enumerate(recs):
    xcc, hwid = hw >> 32, hw & 0xffffffff
    cu = (xcc, (hwid >> 8) & 0xf, (hwid >> 12) & 0x7 if False else (hwid >> 13) & 0x7)      # (XCC, CU_ID, SE_ID) of HW_ID on gfx9
    by_cu[(xcc, hwid & 0x7ff0)].append(((a - t0) / 100, (b - t0) / 100, qt >> 32, qt & 0xffffffff))
loads = sorted((sum(x[3] for x in v), len(v), k) for k, v in by_cu.items())
print(f"{len(by_cu)} distinct (XCC, HW_ID & 0x7ff0) slots; tiles per slot: min {loads[0][0]} median {loads[len(loads) // 2][0]} max {loads[-1][0]}; workgroups per slot: {collections.Counter(l[1] for l in loads)}")
for k in (loads[-1][2], loads[0][2], loads[len(loads) // 2][2]):
    print("  slot", k, " ".join(f"[qb {q} tiles {n}: {a:.1f}-{b:.1f} us]" for a, b, q, n in sorted(by_cu[k])))
durs = collections.defaultdict(list)
for a, b, hw, qt in recs:
    durs[qt & 0xffffffff].append((b - a) / 100)
print("duration by tile count:", " ".join(f"{n}:{sum(v) / len(v):.1f}us" for n, v in sorted(durs.items())))
