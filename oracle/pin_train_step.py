#!/usr/bin/env python
"""Pin oracle/train_oracle.py (Stage-1 loss, projector gradients, AdamW update) against the reference itself.

TEST INFRASTRUCTURE, build container only (imports /root/reference like oracle/pin_against_reference.py, whose model builder and
weight loader it reuses).  The reference model gets the hash-generated weights, everything is frozen except
`model.get_model().mm_projector` (ref:vis_zephyr/train/train.py:817-829), `model(input_ids, attention_mask, labels, images).loss`
is back-propagated, and loss + every projector gradient are compared with the oracle's autograd through its own restatement; then one
torch.optim.AdamW step on three gradient tensors is compared with `adamw_step`.  Writes tests/golden/stage1_step.npz
(loss, per-parameter gradient norms, subsampled slices of six gradients, updated-parameter slices) + stage1_step.json.

    python oracle/pin_train_step.py [--out tests/golden]
"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pin_against_reference as PIN                       # noqa: E402  (sets the import paths: `vis_zephyr` -> the reference)

import numpy as np                                         # noqa: E402
import torch                                               # noqa: E402

from vz_hip import synth                                   # noqa: E402
from oracle import vz_oracle as O, train_oracle as T       # noqa: E402


def batch(cfg):
    """the padded batch of 2 of the forward pin (case B): 2 + 1 tiles, unequal lengths, labels = ids with the image sentinel ignored"""
    tb0, tb1 = synth.synth_tiles(2, seed=3), synth.synth_tiles(1, seed=4)
    ids = torch.full((2, 20), 2, dtype=torch.long)
    ids[0] = synth.synth_ids(20, cfg.vocab, image_pos=1, seed=5)
    ids[1, :13] = synth.synth_ids(13, cfg.vocab, image_pos=7, seed=6)
    mask = torch.zeros(2, 20, dtype=torch.long)
    mask[0] = 1
    mask[1, :13] = 1
    lab = ids.clone()
    lab[ids == O.IMAGE_TOKEN_INDEX] = O.IGNORE_INDEX
    lab[mask == 0] = O.IGNORE_INDEX
    return ids, mask, lab, [tb0, tb1]


def long_batch(cfg):
    """one Stage-1 sample at the length the reference trains up to (ref:script/pretrain.sh:44 --model_max_length 2048 caps it): 5 anyres
    tiles + a 900-id caption -> S = 899 + 160 = 1059 spliced rows; the attention backward of the frozen Zephyr layers runs at S > 1024"""
    tiles = synth.synth_tiles(5, seed=13)
    ids = synth.synth_ids(900, cfg.vocab, image_pos=1, seed=15).unsqueeze(0)
    mask = torch.ones_like(ids)
    lab = ids.clone()
    lab[ids == O.IMAGE_TOKEN_INDEX] = O.IGNORE_INDEX
    lab[:, :2] = O.IGNORE_INDEX
    return ids, mask, lab, [tiles]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(PIN.REPO, "tests", "golden"))
    ap.add_argument("--long", action="store_true", help="the S = 1059 sample (long_batch) -> stage1_long.npz instead of the padded batch of 2")
    args = ap.parse_args()
    cfg = synth.ArchConfig(n_layers=2)
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        model = PIN.build_reference(cfg, tmp)
    sd = PIN.load_synth(model, cfg, 0)
    print(f"[pin-train] reference ready {time.time() - t0:.0f}s", flush=True)
    ids, mask, lab, images = long_batch(cfg) if args.long else batch(cfg)
    tag = "stage1_long" if args.long else "stage1_step"

    # ---- the reference's Stage-1 step ----
    model.requires_grad_(False)
    for p in model.get_model().mm_projector.parameters():
        p.requires_grad_(True)
    out = model(input_ids=ids, attention_mask=mask, labels=lab, images=images)
    out.loss.backward()
    ref_grads = {"model.mm_projector." + n: p.grad.detach().clone() for n, p in model.get_model().mm_projector.named_parameters()}
    ref_loss = float(out.loss)
    print(f"[pin-train] reference loss {ref_loss:.6f}, {len(ref_grads)} gradients {time.time() - t0:.0f}s", flush=True)
    model.zero_grad(set_to_none=True)

    # ---- the oracle ----
    loss, grads = T.stage1_grads(cfg, sd, ids, mask, lab, images)
    print(f"[pin-train] oracle loss {float(loss):.6f} {time.time() - t0:.0f}s", flush=True)
    report = {"loss_ref": ref_loss, "loss_oracle": float(loss), "grads": {}}
    assert abs(float(loss) - ref_loss) <= 2e-6 * abs(ref_loss), (float(loss), ref_loss)
    assert set(grads) == set(ref_grads), sorted(set(grads) ^ set(ref_grads))[:6]
    worst = 0.0
    for k in sorted(grads):
        e = PIN.rel_err(grads[k], ref_grads[k])
        report["grads"][k] = e
        worst = max(worst, e)
        assert e <= 5e-4, f"{k}: gradient rel err {e:.3e}"
    print(f"[pin-train] {len(grads)} projector gradients match, worst rel err {worst:.3e}", flush=True)

    # ---- AdamW ----
    fx = {"loss": np.float64(ref_loss)}
    names = sorted(grads)
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array([float(ref_grads[k].double().norm()) for k in names])
    picks = [k for k in names if k.endswith(("learned_queries", "blocks.0.cross_attention.in_proj_weight", "blocks.7.feed_forward.2.weight",
                                             "ln_vision.weight", "pre_norm.weight", "blocks.3.self_attention.out_proj.bias", "final_norm.bias"))][:6]
    if len(picks) < 3:
        picks = names[:3] + names[-3:]
    for k in picks:
        s, stride = PIN.sub(ref_grads[k])
        fx[f"grad.{k}.sub"] = s
        fx[f"grad.{k}.stride"] = np.int64(stride)
    for k in picks[:3]:
        p0 = sd[k].detach().double().clone()                  # float64: a 2e-5 step on fp32 weights is at the rounding floor
        gk = ref_grads[k].double()
        par = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([par], lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        m = torch.zeros_like(p0)
        v = torch.zeros_like(p0)
        mine = p0.clone()
        for t in (1, 2):
            par.grad = gk * float(t)
            opt.step()
            mine, m, v = T.adamw_step(mine, gk * float(t), m, v, t, 2e-5)
        e = PIN.rel_err(mine - p0, par.detach() - p0)
        report.setdefault("adamw", {})[k] = e
        assert e <= 1e-9, f"AdamW update of {k}: rel err {e:.3e}"
        s, stride = PIN.sub((par.detach() - p0) * 1e5)
        fx[f"adamw_delta_x1e5.{k}.sub"] = s
    print(f"[pin-train] AdamW (2 steps) matches torch.optim.AdamW on {len(picks[:3])} tensors", flush=True)
    assert abs(T.lr_at(0, 1000) - 0.0) < 1e-12 and abs(T.lr_at(30, 1000) - 2e-3) < 1e-12 and T.lr_at(1000, 1000) < 1e-12
    os.makedirs(args.out, exist_ok=True)
    np.savez_compressed(os.path.join(args.out, tag + ".npz"), **fx)
    with open(os.path.join(args.out, tag + ".json"), "w") as f:
        json.dump(report, f, indent=1)
    print(f"[pin-train] OK - fixtures written ({time.time() - t0:.0f}s)", flush=True)


if __name__ == "__main__":
    main()
