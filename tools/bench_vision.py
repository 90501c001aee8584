#!/usr/bin/env python
"""CLIP tower + Q-Former of the 5-tile bench workload, timed per stage with HIP events (one process, interleaved A/B of the
knobs that only touch these stages: 23 = key split of the Q-Former cross-attention, 24 = split-K cap of its 160-row linears,
25 = cross-attention K|V projections of all blocks as one GEMM, 26 = K slices for tile GEMM grids that leave a CU one workgroup).

    python tools/bench_vision.py [tiles]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import torch  # noqa: E402

from vz_hip import binding as B  # noqa: E402
from vz_hip.engine import Engine  # noqa: E402
from vz_hip.synth import ArchConfig  # noqa: E402


def timed(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    cfg = ArchConfig(n_layers=1)
    eng = Engine(cfg, max_batch=1, max_ctx=64, max_tiles=max(8, T))
    eng.load_synthetic(0)
    g = torch.Generator().manual_seed(1)
    tiles = torch.randn(T, 3, cfg.clip_image, cfg.clip_image, generator=g).cuda().bfloat16()
    feats = eng.clip_fused_features(tiles)
    ts = [0] * T
    print(f"tiles {T}", flush=True)
    for rnd in range(3):
        row = []
        for v in (0, 1):
            B.check(B.lib().vz_tune_set(26, v))
            row.append(f"clip {timed(lambda: eng.clip_fused_features(tiles)):6.3f} ms ({'K slices for lone-workgroup grids' if v else 'whole-K tiles only'})")
        B.check(B.lib().vz_tune_set(26, 1))
        for name, knobs in (("qformer base (no key split, 4 slices, K|V per block)", ((23, 1), (24, 4), (25, 0))), ("key split", ((23, 0), (24, 4), (25, 0))),
                            ("+ 8 slices", ((23, 0), (24, 8), (25, 0))), ("+ K|V of all blocks in one GEMM", ((23, 0), (24, 8), (25, 1)))):
            for k, v in knobs:
                B.check(B.lib().vz_tune_set(k, v))
            row.append(f"{name} {timed(lambda: eng.qformer(feats, None, ts)):6.3f} ms")
        B.check(B.lib().vz_tune_set(23, 0)); B.check(B.lib().vz_tune_set(24, 8)); B.check(B.lib().vz_tune_set(25, 1))
        print("   ".join(row), flush=True)


if __name__ == "__main__":
    main()
