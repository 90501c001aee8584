#!/usr/bin/env python
"""TEST INFRASTRUCTURE (not part of the product path).  The bf16 band of the Stage-1 training step's gradients: the oracle
(oracle/train_oracle.py) on the fixture batch of oracle/pin_train_step.py in fp32 and under its BF16 policy (forward AND backward rounded at
the HIP path's store points) -> per tensor e_or = ||g_bf16 - g_fp32|| / ||g_fp32||, written to tests/golden/stage1_step_band.npz so that
tests/test_train_gpu.py does not recompute the BF16-policy backward (about a minute of host time) on every run.  The fp32 oracle gradients the
HIP gradients are compared with are still computed by the test itself (they are gigabytes).
    python oracle/band_train_step.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
from oracle import pin_train_step, train_oracle as T, vz_oracle as O  # noqa: E402
from vz_hip import synth  # noqa: E402


def main():
    t0 = time.time()
    cfg = synth.ArchConfig(n_layers=2)
    sd = {k: v.cpu() for k, v in synth.iter_state_dict(cfg, 0, device="cpu")}
    ids, mask, lab, images = pin_train_step.batch(cfg)
    loss32, g32 = T.stage1_grads(cfg, sd, ids, mask, lab, images)
    print(f"[band-train] fp32 oracle loss {float(loss32):.6f} {time.time() - t0:.0f}s", flush=True)
    loss16, g16 = T.stage1_grads(cfg, sd, ids, mask, lab, images, P=O.BF16)
    print(f"[band-train] bf16-policy oracle loss {float(loss16):.6f} {time.time() - t0:.0f}s", flush=True)
    names = sorted(g32)
    band = []
    for n in names:
        a, b = g16[n].double().reshape(-1), g32[n].double().reshape(-1)
        band.append(float((a - b).norm() / b.norm().clamp_min(1e-30)))
    out = os.path.join(REPO, "tests", "golden", "stage1_step_band.npz")
    np.savez(out, names=np.array(names), band=np.array(band, dtype=np.float64), loss_fp32=np.float64(float(loss32)), loss_bf16=np.float64(float(loss16)))
    print(f"[band-train] {len(names)} tensors, band {min(band):.3e} .. {max(band):.3e} -> {out}", flush=True)


if __name__ == "__main__":
    main()
