"""oracle/train_oracle.py (Stage-1 loss / projector gradients / AdamW - the oracle a later round's HIP backward is held to)
against the fixtures captured from the reference's own `loss.backward()` and `torch.optim.AdamW` (oracle/pin_train_step.py,
tests/golden/stage1_step.npz).  The optimiser and schedule checks are instant; recomputing the 165 gradients through the
full-width Q-Former takes minutes of CPU and runs only with VZ_TRAIN_ORACLE_FULL=1 (the pin script does it on every run)."""
import json
import math
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    path = os.path.join(GOLD, "stage1_step.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/stage1_step.npz not generated")
    return np.load(path, allow_pickle=False), json.load(open(os.path.join(GOLD, "stage1_step.json")))


def test_pin_report_is_tight(gold):
    _, rep = gold
    assert abs(rep["loss_oracle"] - rep["loss_ref"]) <= 2e-6 * abs(rep["loss_ref"])
    assert len(rep["grads"]) == 165                       # every mm_projector parameter (8 blocks + queries + 2 norms)
    assert max(rep["grads"].values()) <= 5e-4
    assert max(rep["adamw"].values()) <= 1e-9


def test_adamw_matches_torch():
    from oracle import train_oracle as T
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(257, 33, generator=g, dtype=torch.float64) * 0.02
    par = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([par], lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    mine, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    for t in range(1, 6):
        grad = torch.randn(p0.shape, generator=g, dtype=torch.float64) * (10.0 ** (t - 3))
        par.grad = grad.clone()
        opt.step()
        mine, m, v = T.adamw_step(mine, grad, m, v, t, 2e-5, weight_decay=0.01)
    d_ref, d_mine = par.detach() - p0, mine - p0
    assert float((d_mine - d_ref).abs().max() / d_ref.abs().max()) <= 1e-10


def test_schedule_matches_hf():
    from oracle import train_oracle as T
    from transformers import get_cosine_schedule_with_warmup
    total = 1000
    par = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([par], lr=2e-3)            # --mm_projector_lr (ref:script/pretrain.sh:16): the projector groups' base rate
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=math.ceil(total * 0.03), num_training_steps=total)
    for step in range(total):
        assert abs(sch.get_last_lr()[0] - T.lr_at(step, total)) <= 1e-12, step
        from vz_hip import train as HT
        assert HT.lr_at(step, total) == T.lr_at(step, total)          # the product-side schedule is the same function
        opt.step()
        sch.step()


@pytest.mark.skipif(os.environ.get("VZ_TRAIN_ORACLE_FULL", "0") != "1", reason="minutes of CPU: set VZ_TRAIN_ORACLE_FULL=1")
def test_gradients_against_reference_fixtures(gold):
    from oracle import train_oracle as T, vz_oracle as O
    from vz_hip import synth
    fx, _ = gold
    cfg = synth.ArchConfig(n_layers=2)
    sd = synth.state_dict(cfg, 0)
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
    loss, grads = T.stage1_grads(cfg, sd, ids, mask, lab, [tb0, tb1])
    assert abs(float(loss) - float(fx["loss"])) <= 2e-6 * abs(float(fx["loss"]))
    names = [str(n) for n in fx["grad_names"]]
    assert sorted(grads) == names
    norms = np.array([float(grads[k].double().norm()) for k in names])
    assert np.allclose(norms, fx["grad_norms"], rtol=2e-4, atol=1e-12)
    for key in fx.files:
        if key.startswith("grad.") and key.endswith(".sub"):
            k = key[len("grad."):-len(".sub")]
            stride = int(fx[f"grad.{k}.stride"])
            mine = grads[k].reshape(-1)[::stride][:4096].double().numpy()
            ref = fx[key].astype(np.float64)
            assert np.abs(mine - ref).max() <= 5e-4 * np.abs(ref).max(), k


def test_band_fixture_belongs_to_the_step_fixture():
    """tests/golden/stage1_step_band.npz (oracle/band_train_step.py: the BF16-policy oracle's per-tensor distance from its own fp32 gradients) was
    made on the SAME batch and weights as stage1_step.npz: same 165 tensors, the same loss (the reference's, to 1e-5), bands of bf16 size."""
    import os
    import numpy as np
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(golden, "stage1_step.npz"), allow_pickle=False)
    b = np.load(os.path.join(golden, "stage1_step_band.npz"), allow_pickle=False)
    assert sorted(str(n) for n in b["names"]) == sorted(str(n) for n in g["grad_names"]) and len(b["names"]) == 165
    assert abs(float(b["loss_fp32"]) - float(g["loss"])) <= 1e-5 * float(g["loss"])
    assert abs(float(b["loss_bf16"]) - float(g["loss"])) <= 1e-3 * float(g["loss"])
    assert 5e-3 < float(b["band"].min()) and float(b["band"].max()) < 8e-2
