#!/usr/bin/env python
"""A handful of launches of the prefill GEMMs for a PMC pass (rocprofv3 --pmc ... --kernel-include-regex gemm): the four
Zephyr shapes at S=2048 through the production dispatch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import torch  # noqa: E402

from vz_hip import binding as B  # noqa: E402

for name, M, N, K, act in [("qkv", 2048, 6144, 4096, 0), ("o", 2048, 4096, 4096, 0), ("gate-up", 2048, 28672, 4096, 3), ("down", 2048, 4096, 14336, 0)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    for _ in range(6):
        B.linear(x, w, act=act)
    torch.cuda.synchronize()
print("done")
