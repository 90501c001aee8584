"""The CPU oracle against the golden vectors captured from the reference itself
(oracle/pin_against_reference.py): fp32, <= 1e-5 relative.  Inputs and weights regenerate from seeds."""
import numpy as np
import pytest
import torch

from util import load_golden, sub


@pytest.fixture(scope="module")
def env():
    from oracle import vz_oracle as O
    from vz_hip import synth
    cfg = synth.ArchConfig(n_layers=2)
    sd = synth.state_dict(cfg, 0)
    return cfg, sd, O, synth, load_golden(2)


def _chk(g, name, t, tol=2e-5):
    ref = torch.from_numpy(g[name + ".sub"]).double()
    mine = sub(t, int(g[name + ".stride"])).double()
    assert tuple(t.shape) == tuple(g[name + ".shape"].tolist())
    err = float((mine - ref).abs().max() / ref.abs().max())
    assert err <= tol, f"{name}: {err:.3e}"
    assert abs(float(t.double().sum()) - float(g[name + ".sum"])) <= 1e-4 * float(g[name + ".abssum"])


def test_case_a_all_stages(env):
    cfg, sd, O, S, g = env
    tiles = S.synth_tiles(3, seed=1)
    ids = S.synth_ids(32, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    hs = O.clip_hidden_states(cfg, sd, tiles)
    for i in (0, 1, 4, 12, 24):
        _chk(g, f"A.clip.hs{i}", hs[i])
    fused = O.fusion(cfg, hs)
    _chk(g, "A.fused", fused)
    te = O.embed_tokens(sd, ids[0][ids[0] != -200], O.FP32).unsqueeze(0).expand(3, -1, -1)
    out, blocks = O.qformer(cfg, sd, fused, te, return_blocks=True)
    for i in (0, 1, 7):
        _chk(g, f"A.qformer.block{i}", blocks[i])
    _chk(g, "A.qformer.out", out)
    r = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles],
                                               encode_fn=lambda im, t: out)
    assert r[0] is None and r[1] is None and r[2] is None and r[5] is None
    _chk(g, "A.splice.embeds", r[4])
    lo, _ = O.llm_forward(cfg, sd, r[4])
    _chk(g, "A.logits", lo, tol=5e-5)
    assert np.allclose(lo[0, -1].numpy(), g["A.logits.last"], atol=2e-4)
    gen, glog = O.greedy_generate(cfg, sd, r[4], 6, return_logits=True)
    assert gen.tolist() == g["A.generate.ids"].tolist()
    assert np.allclose(glog[0, :, ::37].numpy(), g["A.generate.step_logits.sub"], atol=3e-4)


def test_case_d_cls_patch(env):
    """mm_vision_select_feature = 'cls_patch' (ref vision_encoder.py:66-73): 577 tokens per tile."""
    cfg, sd, O, S, g = env
    tiles = S.synth_tiles(2, seed=8)
    ids = S.synth_ids(32, cfg.vocab, image_pos=5, seed=2)
    te = O.embed_tokens(sd, ids[ids != -200], O.FP32).unsqueeze(0).expand(2, -1, -1)
    fused = O.clip_tower(cfg, sd, tiles, select_feature="cls_patch")
    _chk(g, "D.fused.cls_patch", fused)
    _chk(g, "D.encode_images.cls_patch", O.qformer(cfg, sd, fused, te))
    with pytest.raises(ValueError):
        O.fusion(cfg, [fused] * 25, select_feature="cls")


def test_case_c_text_only_generate(env):
    cfg, sd, O, S, g = env
    ids = S.synth_ids(9, cfg.vocab, image_pos=-1, seed=7).unsqueeze(0)
    assert O.generate(cfg, sd, ids, None, 4).tolist() == g["C.generate.ids"].tolist()


def test_bf16_oracle_stays_in_band_of_fp32(env):
    """the bf16-rounding oracle is the GPU path's yardstick; it must itself stay close to the fp32 reference."""
    cfg, sd, O, S, g = env
    ids = S.synth_ids(24, cfg.vocab, image_pos=-1, seed=8).unsqueeze(0)
    e32 = O.embed_tokens(sd, ids, O.FP32)
    l32, _ = O.llm_forward(cfg, sd, e32)
    l16, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids, O.BF16), P=O.BF16)
    rel = float((l16 - l32).norm() / l32.norm())
    assert rel < 2e-2, rel


def test_depth32_fixture_is_complete_and_consistent():
    """tests/golden/pin_l32.npz (oracle/pin_against_reference.py --llm-layers 32 --deep): the 32-layer fixtures the GPU tests
    in test_depth32_gpu.py compare against.  Re-running the 32-layer oracle needs 37 GB and minutes, so here only what can be
    checked from the data itself: every key is present, the report's worst stage error is below the pin's bound, and the three
    precisions order as they must (weight rounding < bf16 band + weight rounding)."""
    import json
    import os

    import numpy as np
    from util import GOLDEN, load_golden
    g = load_golden(32)
    meta = json.load(open(os.path.join(GOLDEN, "pin_l32.json")))
    assert meta["llm_layers"] == 32 and max(v for k, v in meta["report"].items() if "rounding" not in k and "divergence" not in k) <= 5e-5
    for k in ("A.logits.sub", "A.logits.last", "A16.logits.last", "A.generate.ids", "A16.generate.ids", "E.generate.ids", "E16.generate.ids",
              "E.step_logits.s64", "E16.step_logits.s64", "E16.logits.last", "E16.step_top2.vals", "E.bf16_oracle.step_logits.s64",
              "E.w16_on_fp32_ids.step_logits.s64", "B.loss", "C.generate.ids"):
        assert k in g, k
    assert g["E.generate.ids"].shape == g["E16.generate.ids"].shape == (1, 128) and g["E16.step_logits.s64"].shape == (128, 500)
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))  # noqa: E731
    e_w = rel(g["E.w16_on_fp32_ids.step_logits.s64"], g["E.step_logits.s64"])
    e_a = rel(g["E.bf16_oracle.step_logits.s64"], g["E.w16_on_fp32_ids.step_logits.s64"])
    e_t = rel(g["E.bf16_oracle.step_logits.s64"], g["E.step_logits.s64"])
    assert 1e-3 < e_w < 0.1 and 1e-3 < e_a < 0.1 and e_t < e_w + e_a + 1e-3
    # the reference's own fp32 and W16 runs agree on the first token and on argmax = id at every step of their own runs
    assert int(np.argmax(g["E16.logits.last"])) == int(g["E16.generate.ids"][0, 0])
    assert (g["E16.step_top2.ids"][:, 0] == g["E16.generate.ids"][0]).all()
