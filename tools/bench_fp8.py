#!/usr/bin/env python
"""fp8 MFMA prefill linears (gemm_fp8.hip) against the bf16 tile GEMMs on the Zephyr prefill shapes (M = 2048), and the activation quantiser.
    python tools/bench_fp8.py [M]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import torch  # noqa: E402

from vz_hip import binding as B, quant  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = "cuda:0"


def timed(fn, n=10):
    fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, N, K, act in (("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate-up", 28672, 4096, 3), ("down", 4096, 14336, 0)):
    x = torch.randn(M, K, device=dev).bfloat16()
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(3)]
    q = [quant.quantize_rows(w) for w in ws]
    x8, xs = B.quant_rows_fp8(x)
    res = torch.randn(M, N // 2 if act == 3 else N, device=dev).bfloat16()
    fl = 2.0 * M * N * K
    t_bf = min(timed(lambda i: B.linear(x, ws[i % 3], residual=res, act=act)) for _ in range(3))
    t_f8 = min(timed(lambda i: B.linear_fp8_mfma(x8, xs, q[i % 3][0], q[i % 3][1], residual=res, act=act)) for _ in range(3))
    t_q = min(timed(lambda i: B.quant_rows_fp8(x)) for _ in range(3))
    print(f"{name:8s} M{M} N{N} K{K}: bf16 dispatch {t_bf:7.1f} us {fl / t_bf / 1e6:7.1f} TF | fp8 mfma {t_f8:7.1f} us {fl / t_f8 / 1e6:7.1f} TF | quantise x {t_q:5.1f} us",
          flush=True)
