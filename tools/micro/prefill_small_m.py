#!/usr/bin/env python
"""The Zephyr prefill linears at SHORT prompts (M = 64..512 rows: the CLI's typical request is 5 tiles x 32 + ~40 ids = 199 rows): time per GEMM by route
against the weight-stream floor.   python tools/micro/prefill_small_m.py [M ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd")); sys.path.insert(0, REPO)
import torch
from vz_hip import binding as B
dev = "cuda:0"
def timed(fn, n):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
Ms = [int(a) for a in sys.argv[1:]] or [199, 512]
for M in Ms:
    tot = {}
    for name, N, K, act in (("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 28672, 4096, 3), ("down", 4096, 14336, 0)):
        x = torch.randn(M, K, device=dev).bfloat16()
        ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(3)]
        row = []
        for label, kw in (("dispatch", {}), ("128", {"impl": 0}), ("256sk", {"impl": 2})):
            try:
                us = min(timed(lambda i: B.linear(x, ws[i % 3], act=act, **kw), 20) for _ in range(2))
            except Exception as ex:
                us = float("nan")
            row.append(f"{label}: {us:7.1f} us")
            tot[label] = tot.get(label, 0.0) + us
        fl = max(N * K * 2 / 6.3e6, 2.0 * M * N * K / 1.0e9)
        print(f"M{M} {name:8s} N{N} K{K}: " + "  ".join(row) + f"   floor {fl:6.1f} us", flush=True)
    print(f"M{M} per layer: " + "  ".join(f"{k}: {v:7.1f} us" for k, v in tot.items()), flush=True)
