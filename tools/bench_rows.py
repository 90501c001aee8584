#!/usr/bin/env python
"""17..64-row decode linears: the MFMA weight stream (gemm_skinny.hip, impl 3) against the tile GEMMs (128^2 impl 0 / 256^2 impl 2) on the
four Zephyr decode shapes.  python tools/bench_rows.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd")); sys.path.insert(0, REPO)
import torch
from vz_hip import binding as B
B.load_library()
shapes = {"qkv": (6144, 4096, 0), "o": (4096, 4096, 0), "gate-up": (28672, 4096, 3), "down": (4096, 14336, 0)}
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for M in (17, 32, 48, 64):
    line = f"M={M:3d}"
    for name, (N, K, act) in shapes.items():
        x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        res = {}
        for impl in (3, 0, 4):
            try:
                res[impl] = t(lambda: B.linear(x, w, act=act, impl=impl))
            except Exception as ex:
                res[impl] = float("nan")
        gb = N * K * 2 / 1e3
        line += f" | {name}: skinny {res[3]:6.1f} us ({gb / res[3]:5.0f} GB/s)  g128 {res[0]:6.1f} ({gb / res[0]:5.2f})  g128-sk {res[4]:6.1f} ({gb / res[4]:5.2f})"
    print(line, flush=True)
