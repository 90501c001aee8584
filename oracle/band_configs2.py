#!/usr/bin/env python
"""The bf16 activation band of BASELINE configs[2] (bench.py's request) at 32 decoder layers.  TEST INFRASTRUCTURE.

tests/golden/pin_l32_c2.npz holds what the imported reference computes for that request on bf16-rounded matrices (W16:
`F16.*`, written by oracle/pin_against_reference.py --configs2, which also checks this oracle against it to 5e-5).  This script
adds the other side of the band: the BF16-policy oracle (oracle/vz_oracle.py: a bf16 rounding at every point where the HIP
path stores bf16 to HBM) on the SAME request, teacher-forced on the reference's own 16 greedy ids -

    C2.bf16_oracle.logits.last        [V]       last prefill row
    C2.bf16_oracle.step_logits.s64    [16, 500] the 16 step logits, 1-in-64 of the vocabulary (as F16.step_logits.s64)
    C2.w16_oracle.step_logits.s64     [16, 500] the W16 oracle on the same ids (must reproduce F16.step_logits.s64: asserted)

so that tests/test_depth32_gpu.py can hold the HIP path to  ||HIP - F16|| <= 1.15 ||BF16 - F16||  on the headline request
itself instead of borrowing configs[1]'s band.  No reference import: the oracle is pinned already; needs ~40 GB of host
memory and a few minutes of CPU.  Output: tests/golden/pin_l32_c2_band.npz (+ .json).

    python oracle/band_configs2.py [--llm-layers 32]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)

import numpy as np   # noqa: E402
import torch         # noqa: E402

from oracle import vz_oracle as O   # noqa: E402
from vz_hip import synth            # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm-layers", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    t0 = time.time()
    cfg = synth.ArchConfig(n_layers=args.llm_layers)
    g = np.load(os.path.join(args.out, f"pin_l{args.llm_layers}_c2.npz"))
    ref_ids = torch.from_numpy(g["F16.generate.ids"][0])
    n_new = int(ref_ids.numel())
    sd = {k: v for k, v in synth.iter_state_dict(cfg, 0)}
    print(f"[band] weights generated {time.time() - t0:.0f}s", flush=True)
    tiles = synth.synth_tiles(5, seed=1)
    ids = synth.synth_ids(1889, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    fx, rep = {}, {}
    for tag, P in (("w16", O.W16), ("bf16", O.BF16)):
        emb = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=P)[4]
        S = emb.shape[1]
        assert S == 2048
        print(f"[band] {tag}: vision + splice done {time.time() - t0:.0f}s", flush=True)
        full = torch.cat([emb, O.embed_tokens(sd, ref_ids[:-1], P).unsqueeze(0)], 1)
        _, _, hfin = O.llm_forward(cfg, sd, full, P=P, last_only=True, return_hidden=True)
        step = (hfin[0, S - 1:] @ P.w(sd["lm_head.weight"]).t()).float()
        assert step.shape[0] == n_new
        fx[f"C2.{tag}_oracle.step_logits.s64"] = step[:, ::64].numpy().astype(np.float32)
        fx[f"C2.{tag}_oracle.logits.last"] = step[0].numpy().astype(np.float32)
        print(f"[band] {tag}: decoder done {time.time() - t0:.0f}s", flush=True)
        del emb, full, hfin, step
    e_pin = rel(fx["C2.w16_oracle.step_logits.s64"], g["F16.step_logits.s64"])
    rep["w16_oracle_vs_reference_fixture_rel_l2"] = e_pin
    assert e_pin <= 1e-4, f"the W16 oracle does not reproduce the reference's fixture: {e_pin:.3e}"
    rep["band_last_row"] = rel(fx["C2.bf16_oracle.logits.last"], g["F16.logits.last"])
    rep["band_16_steps"] = rel(fx["C2.bf16_oracle.step_logits.s64"], g["F16.step_logits.s64"])
    rep["band_per_step"] = [rel(fx["C2.bf16_oracle.step_logits.s64"][t], g["F16.step_logits.s64"][t]) for t in range(n_new)]
    rep["llm_layers"] = args.llm_layers
    rep["torch"] = torch.__version__
    rep["note"] = ("BF16-policy oracle vs the W16 reference fixture (pin_l32_c2.npz) on BASELINE configs[2], teacher-forced on the "
                   "reference's 16 greedy ids; the W16 oracle column must equal the fixture (asserted)")
    del fx["C2.w16_oracle.logits.last"]
    np.savez_compressed(os.path.join(args.out, f"pin_l{args.llm_layers}_c2_band.npz"), **fx)
    with open(os.path.join(args.out, f"pin_l{args.llm_layers}_c2_band.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(f"[band] OK {json.dumps({k: v for k, v in rep.items() if k != 'band_per_step'})} ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
