import os, sys, time
sys.path.insert(0, "vision-zephyr_amd"); sys.path.insert(0, ".")
import torch
from vz_hip import binding as B
dev = "cuda:0"
def timed(fn, n):
    fn(0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for (M, N, K) in ((128, 4096, 4096), (160, 4096, 4096), (128, 12288, 4096), (160, 12288, 4096), (128, 4096, 8192), (160, 4096, 8192), (32, 4096, 4096), (64, 4096, 4096)):
    x = torch.randn(M, K, device=dev).bfloat16()
    ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(8)]      # 8 x 33.5 MB: rotates through more than the L2s hold
    row = []
    for sl in (2, 4, 8, 16):
        B.check(B.lib().vz_tune_set(24, sl))
        us = min(timed(lambda i: B.linear(x, ws[i % 8]), 40) for _ in range(3))
        row.append(f"slices<={sl}: {us:6.1f} us")
    B.check(B.lib().vz_tune_set(24, 8))
    print(f"M{M} N{N} K{K}: " + "  ".join(row) + f"   (weights {N * K * 2 / 1e6:.1f} MB: {N * K * 2 / 6.3e6:.1f} us at 6.3 TB/s)", flush=True)
