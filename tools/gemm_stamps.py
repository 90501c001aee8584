#!/usr/bin/env python
"""In-kernel phase timeline of the 256x256 GEMM (stream-K on) for one shape: where a workgroup's time goes.

    python tools/gemm_stamps.py M N K [streamk]
Stamps are s_memrealtime (100 MHz); printed in microseconds relative to the first workgroup's start.  Launches are repeated
for ~1 s first so that the clock the chip settles at under this load is the one measured."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vz_hip import binding as B  # noqa: E402

M, N, K = (int(a) for a in sys.argv[1:4])
sk = int(sys.argv[4]) if len(sys.argv) > 4 else 1
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
if os.environ.get("VZ_ZERO", "0") == "1":      # all-zero operands: the clock the chip holds when the data toggles nothing (guide: DVFS give-back item 1)
    x.zero_(); w.zero_()
B.check(B.lib().vz_tune_set(4, sk))
for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):       # experiments: "knob=value,..." (34=0: direct fragment-shaped epilogue stores)
    _k, _v = (int(t) for t in kv.split("="))
    B.check(B.lib().vz_tune_set(_k, _v))
ACT = int(os.environ.get("VZ_ACT", "0"))
import time
t_end = time.time() + 1.0
while time.time() < t_end:                 # let DVFS settle under this kernel's load
    for _ in range(50):
        B.linear(x, w, impl=2, act=ACT)
    torch.cuda.synchronize()
B.check(B.lib().vz_tune_set(6, 1))
for _ in range(3):
    B.linear(x, w, impl=2, act=ACT)
torch.cuda.synchronize()
buf = (C.c_longlong * (4096 * 16))()
n = C.c_int(0)
B.check(B.lib().vz_prof_gemm_stamps(buf, 4096, C.byref(n)))
B.check(B.lib().vz_tune_set(6, 0))
st = np.ctypeslib.as_array(buf).reshape(4096, 16)[: n.value].astype(np.int64)
t0 = st[:, 0].min()
us = lambda a: (a - t0) / 100.0
print(f"shape {M}x{N}x{K} streamk={sk}: {n.value} workgroups; start spread {us(st[:, 0]).max():.1f} us; "
      f"kernel span {us(st[:, 1:14][st[:, 1:14] > 0].max()):.1f} us")
# shader clock inside the main loop of the first K-slice: delta s_memtime / delta s_memrealtime x 100 MHz (guide, DVFS item 6)
ok = (st[:, 15] > st[:, 14]) & (st[:, 2] > st[:, 1])
if ok.any():
    ghz = (st[ok, 15] - st[ok, 14]) / (st[ok, 2] - st[ok, 1]) * 0.1
    print(f"  in-kernel shader clock during the main loop: median {np.median(ghz):.2f} GHz (min {ghz.min():.2f}, max {ghz.max():.2f})")
for seg in range(3):
    b = 1 + seg * 5
    have = st[:, b] > 0
    if not have.any():
        continue
    s = st[have]
    info = s[:, b + 4]
    nks = info >> 32
    fin = (info & 1) == 1
    part = (info & 2) == 2
    prev = s[:, 0] if seg == 0 else s[:, b - 2]
    for label, sel in (("whole tile", ~part), ("slice, parks", part & ~fin), ("slice, finishes", part & fin)):
        if not sel.any():
            continue
        q = s[sel]
        fix = np.where(q[:, b + 2] > 0, q[:, b + 2], q[:, b + 1])
        med = lambda a: float(np.median(a)) / 100.0
        print(f"  segment {seg} {label:16s} n={int(sel.sum()):4d} nk~{int(np.median(nks[sel])):4d}: "
              f"prologue {med(q[:, b] - prev[sel]):6.1f}  loop {med(q[:, b + 1] - q[:, b]):6.1f} "
              f"({med(q[:, b + 1] - q[:, b]) / max(1, float(np.median(nks[sel]))):.2f}/K-tile)  fix-up {med(fix - q[:, b + 1]):6.1f}  "
              f"epilogue {med(q[:, b + 3] - fix):6.1f}  | done at {med(q[:, b + 3] - t0):6.1f} (max {us(q[:, b + 3]).max():6.1f})")
