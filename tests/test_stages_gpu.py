"""Stage- and API-level parity of the MI355X path against the CPU oracle and the golden vectors.

Model under test: the full-width architecture (CLIP-L/14-336, 8-block Q-Former, Zephyr width 4096 /
14336, vocab 32000) with 2 decoder layers and the hash-generated weights of seed 0 - the exact model
the golden vectors in tests/golden/pin_l2.npz were produced on by the imported reference.

Tolerances.  Single kernels are held to one bf16 rounding (tests/test_ops_gpu.py; fp32-output kernels
to 1e-4).  Through a stack of layers two *correct* bf16 implementations decorrelate: every bf16 store
is an independent 2^-9 rounding, so the HIP path and the bf16-rounding oracle differ from each other by
about as much as either differs from the fp32 reference.  The stage tests therefore measure the bf16
BAND itself and hold the HIP path to it:
    e_or  = ||oracle_bf16 - oracle_fp32|| / ||oracle_fp32||     (what ideal bf16 arithmetic costs)
    e_hip = ||hip - oracle_fp32||        / ||oracle_fp32||      must be <= band_tol(e_or)
    e_x   = ||hip - oracle_bf16||        / ||oracle_bf16||      must be <= 2.2 * e_or + 5e-4
i.e. the GPU path may not be measurably worse than bf16 arithmetic done in fp32 on the CPU, and the
fp32 golden vectors of the reference itself bound it in absolute terms (per-test constants).
"""
import numpy as np
import pytest
import torch

from util import band_tol, check_close, errs, load_golden, record, sub

pytestmark = pytest.mark.gpu

PINPOINTS = "[[336, 672], [672, 336], [336, 1008], [1008, 336]]"


@pytest.fixture(scope="module")
def env():
    from oracle import vz_oracle as O
    from vz_hip import synth
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg = synth.ArchConfig(n_layers=2)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.image_aspect_ratio = "anyres"
    hf.mm_grid_pinpoints = PINPOINTS
    hf.mm_hidden_size = 5120
    hf.mm_vision_select_layer = "-2,-5,-8,-11,6"
    hf.mm_vision_select_feature = "patch"
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=2, max_ctx=512, max_tiles=4, max_text=64)
    # the oracle's fp32 weights: generated on the device (bit-identical to the CPU generator), copied to the host
    sd = {k: v.cpu() for k, v in synth.iter_state_dict(cfg, 0, device=model.device)}
    torch.cuda.synchronize()
    return dict(cfg=cfg, model=model, sd=sd, O=O, synth=synth, gold=load_golden(2))


def band(name, hip, o_bf16, o_fp32):
    """hold the HIP result to the measured bf16 band (module docstring)."""
    e_or = errs(o_bf16, o_fp32)[1]
    mx_hip, e_hip = errs(hip, o_fp32)
    mx_x, e_x = errs(hip, o_bf16)
    record(name, e_oracle_bf16_vs_fp32=e_or, e_hip_vs_fp32=e_hip, e_hip_vs_bf16=e_x, max_hip_vs_fp32=mx_hip)
    assert torch.isfinite(hip.float()).all(), f"{name}: non-finite"
    assert e_hip <= band_tol(e_or), f"{name}: hip vs fp32 {e_hip:.3e} outside the bf16 band (oracle-bf16 vs fp32 {e_or:.3e})"
    assert e_x <= 2.2 * e_or + 5e-4, f"{name}: hip vs oracle-bf16 {e_x:.3e} (band {e_or:.3e})"


def _gold_check(env, name, got, max_tol, l2_tol):
    g = env["gold"]
    ref = torch.from_numpy(g[name + ".sub"])
    mine = sub(got.float().cpu(), int(g[name + ".stride"]))
    assert tuple(got.shape) == tuple(g[name + ".shape"].tolist())
    return check_close("golden " + name, mine, ref, max_tol, l2_tol)


def test_weights_match_generator(env):
    """the device-side generator + packer reproduce the CPU generator bit for bit (spot checks)."""
    eng, sd, cfg = env["model"].engine, env["sd"], env["cfg"]
    q = sd["model.layers.1.self_attn.q_proj.weight"].bfloat16()
    assert torch.equal(eng.w["llm.1.qkv.w"][:4096].cpu(), q)
    g = sd["model.layers.0.mlp.gate_proj.weight"].bfloat16()
    u = sd["model.layers.0.mlp.up_proj.weight"].bfloat16()
    gu = eng.w["llm.0.gu.w"].cpu().view(-1, 2, 16, 4096)
    assert torch.equal(gu[:, 0].reshape(-1, 4096), g) and torch.equal(gu[:, 1].reshape(-1, 4096), u)
    cpu = env["synth"].hash_normal("lm_head.weight", (cfg.vocab, cfg.hidden), 0.02, 0)[:64]
    assert torch.equal(cpu, sd["lm_head.weight"][:64])


def test_clip_tower_and_fusion(env):
    O, cfg, sd, eng = env["O"], env["cfg"], env["sd"], env["model"].engine
    tiles = env["synth"].synth_tiles(3, seed=1)
    fused, hid = eng.clip_fused_features(tiles, return_hidden=True)
    hs = O.clip_hidden_states(cfg, sd, tiles, O.BF16)
    hs32 = O.clip_hidden_states(cfg, sd, tiles, O.FP32)
    check_close("clip hs0 vs oracle-bf16", hid[0], hs[0], 8e-3, 1e-3)      # patch GEMM + assemble + LN only
    for i in (1, 12, 24):
        band(f"clip hs{i}", hid[i], hs[i], hs32[i])
    band("fused", fused, O.fusion(cfg, hs, O.BF16), O.fusion(cfg, hs32, O.FP32))
    # against the reference's fp32 outputs: 24 layers of bf16 activations
    _gold_check(env, "A.clip.hs24", hid[24], 5e-2, 1.5e-2)
    _gold_check(env, "A.fused", fused, 5e-2, 1.5e-2)
    env["fused"] = fused
    env["tiles"] = tiles


def test_fusion_rejects_wrong_image_size(env):
    with pytest.raises(ValueError):
        env["model"].engine.clip_fused_features(torch.zeros(1, 3, 224, 224))


def test_qformer(env):
    O, cfg, sd, model = env["O"], env["cfg"], env["sd"], env["model"]
    eng = model.engine
    tiles = env["synth"].synth_tiles(3, seed=1)
    ids = env["synth"].synth_ids(32, cfg.vocab, image_pos=5, seed=2)
    text_ids = ids[ids != -200]
    te = O.embed_tokens(sd, text_ids, O.BF16)
    feats = O.clip_tower(cfg, sd, tiles, O.BF16)                        # same features into both sides
    ref, blocks = O.qformer(cfg, sd, feats, te.unsqueeze(0).expand(3, -1, -1), O.BF16, return_blocks=True)
    ref32 = O.qformer(cfg, sd, feats, te.unsqueeze(0).expand(3, -1, -1), O.FP32)
    # shared form (one sample, three tiles) and the per-tile form must agree with the oracle and with each other
    shared = eng.qformer(feats.bfloat16(), te.bfloat16().unsqueeze(0), [0, 0, 0])
    per_tile = eng.qformer(feats.bfloat16(), te.bfloat16().unsqueeze(0).expand(3, -1, -1).contiguous(), [0, 1, 2])
    band("qformer shared", shared, ref, ref32)
    band("qformer per-tile", per_tile, ref, ref32)
    assert torch.equal(shared, per_tile), "sharing block 0's self-attention across a sample's tiles changed the result"
    no_text = eng.qformer(feats.bfloat16(), None, [0, 0, 0])
    band("qformer no text", no_text, O.qformer(cfg, sd, feats, None, O.BF16), O.qformer(cfg, sd, feats, None, O.FP32))


def _case_b(env):
    """inputs of case B of the golden set: batch of 2, unequal lengths, padding mask, labels, position ids (built from seeds by
    every test that needs them: no test reads another test's leftovers)"""
    S, cfg = env["synth"], env["cfg"]
    tb0, tb1 = S.synth_tiles(2, seed=3), S.synth_tiles(1, seed=4)
    ids = torch.full((2, 20), 2, dtype=torch.long)
    ids[0] = S.synth_ids(20, cfg.vocab, image_pos=1, seed=5)
    ids[1, :13] = S.synth_ids(13, cfg.vocab, image_pos=7, seed=6)
    mask = torch.zeros(2, 20, dtype=torch.long)
    mask[0] = 1
    mask[1, :13] = 1
    pos = torch.arange(20).unsqueeze(0).expand(2, -1).contiguous()
    lab = ids.clone()
    lab[ids == -200] = -100
    return dict(ids=ids, mask=mask, pos=pos, lab=lab, tiles=[tb0, tb1])


def test_splice_api_matches_oracle_batch(env):
    """case B of the golden set: batch of 2, unequal lengths, padding mask, labels, position ids."""
    O, cfg, sd, model, g = env["O"], env["cfg"], env["sd"], env["model"], env["gold"]
    b = _case_b(env)
    ids, mask, pos, lab, (tb0, tb1) = b["ids"], b["mask"], b["pos"], b["lab"], b["tiles"]
    r = model.prepare_inputs_labels_for_multimodal(ids, pos, mask, None, lab, [tb0, tb1], None)
    assert r[0] is None and r[3] is None
    assert np.array_equal(r[1].cpu().numpy(), g["B.position_ids"])
    assert np.array_equal(r[2].cpu().numpy(), g["B.attention_mask"]) and r[2].dtype == mask.dtype
    assert np.array_equal(r[5].cpu().numpy(), g["B.labels"])
    # embeddings: visual rows through the HIP encoder, text rows exact bf16 table rows
    o = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, pos, mask, None, lab, [tb0, tb1], P=O.BF16)
    o32 = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, pos, mask, None, lab, [tb0, tb1], P=O.FP32)
    band("splice embeds", r[4], o[4], o32[4])
    _gold_check(env, "B.splice.embeds", r[4], 5e-2, 1.5e-2)
    text_rows = torch.from_numpy(g["B.labels"]) != -100
    text_rows[:, 0] = True
    # rows holding text embeddings are bit-exact copies of the bf16 table
    tbl = sd["model.embed_tokens.weight"].bfloat16()
    assert torch.equal(r[4][0, 0].cpu(), tbl[ids[0, 0]])
    # early-outs of the reference (:148-149)
    assert model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, None)[4] is None
    assert model.prepare_inputs_labels_for_multimodal(ids[:, :1], None, None, None, None, [tb0, tb1])[4] is None


def test_forward_logits_case_a(env):
    """C1 shape: 3 tiles + 32 ids -> logits [1,127,32000] for all positions."""
    O, cfg, sd, model = env["O"], env["cfg"], env["sd"], env["model"]
    S = env["synth"]
    tiles = S.synth_tiles(3, seed=1)
    ids = S.synth_ids(32, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    out = model(input_ids=ids, images=[tiles])
    assert tuple(out.logits.shape) == (1, 127, cfg.vocab) and out.logits.dtype == torch.float32
    emb = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=O.BF16)[4]
    ref, _ = O.llm_forward(cfg, sd, emb, P=O.BF16)
    emb32 = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=O.FP32)[4]
    ref32, _ = O.llm_forward(cfg, sd, emb32, P=O.FP32)
    band("logits A", out.logits, ref, ref32)
    _gold_check(env, "A.logits", out.logits, 5e-2, 2e-2)
    last = torch.from_numpy(env["gold"]["A.logits.last"])
    mx, l2 = errs(out.logits[0, -1].cpu(), last)
    record("logits A last vs reference fp32", max_err=mx, l2_err=l2)
    assert l2 < 2e-2


def test_forward_batch_with_padding_and_loss(env):
    O, cfg, sd, model = env["O"], env["cfg"], env["sd"], env["model"]
    b = _case_b(env)
    out = model(input_ids=b["ids"], attention_mask=b["mask"], position_ids=b["pos"], labels=b["lab"], images=b["tiles"])
    emb = O.prepare_inputs_labels_for_multimodal(cfg, sd, b["ids"], b["pos"], b["mask"], None, b["lab"], b["tiles"], P=O.BF16)
    ref, _ = O.llm_forward(cfg, sd, emb[4], attention_mask=emb[2], position_ids=emb[1], P=O.BF16)
    valid = emb[2].bool()
    e32 = O.prepare_inputs_labels_for_multimodal(cfg, sd, b["ids"], b["pos"], b["mask"], None, b["lab"], b["tiles"], P=O.FP32)
    r32, _ = O.llm_forward(cfg, sd, e32[4], attention_mask=e32[2], position_ids=e32[1], P=O.FP32)
    band("logits B (valid rows)", out.logits.cpu()[valid], ref[valid], r32[valid])
    _gold_check(env, "B.logits.valid", out.logits.cpu()[valid], 5e-2, 2e-2)
    assert abs(float(out.loss) - float(env["gold"]["B.loss"])) < 0.05 * abs(float(env["gold"]["B.loss"]))


def test_generate_greedy_matches_oracle_and_reference(env):
    O, cfg, sd, model, g = env["O"], env["cfg"], env["sd"], env["model"], env["gold"]
    S = env["synth"]
    tiles = S.synth_tiles(3, seed=1)
    ids = S.synth_ids(32, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    n_new = 6
    got = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, eos_token_id=None,
                         pad_token_id=2, use_cache=True)
    assert got.dtype == torch.long and tuple(got.shape) == (1, n_new)          # new tokens only
    emb = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=O.BF16)[4]
    ref_ids, ref_logits = O.greedy_generate(cfg, sd, emb, n_new, P=O.BF16, return_logits=True)
    # a greedy step may legitimately leave the oracle's ids only at a near-tie, and "near" is MEASURED (round 3; it was a hand-picked 2e-3 of
    # max|logit|): the HIP logits of that step under teacher forcing (equal prefixes up to there) must sit inside the step's own bf16 band, and the
    # oracle's gap between its choice and the HIP choice must be inside what that step's error can move a logit: gap < 4 x rel-L2 error x rms
    # (the criterion of tests/test_depth32_gpu.py for the headline request)
    top2 = ref_logits[0].topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1]) / ref_logits[0].abs().amax(-1)
    first = next((t for t in range(n_new) if int(got[0, t]) != int(ref_ids[0, t])), -1)
    if first >= 0:
        eng = model.engine
        hip_emb = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles])[4]
        S0 = hip_emb.shape[1]
        _, last = eng.prefill(hip_emb, [S0], all_logits=False, last_logits=True)
        hip_t = last[0].float().cpu()
        forced = ref_ids[0].to(torch.int32)
        if first > 0:
            eng.decode_begin(forced[:1], [S0], [S0])
            for t in range(1, first + 1):
                _, lg = eng.decode_steps(1, return_logits=True)
                hip_t = lg[0, 0].float().cpu()
                if t < first:
                    eng.decode_set_row(0, int(forced[t]), S0 + t, S0 + t)
        ref_t = ref_logits[0, first].float()
        ext32 = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=O.FP32)[4]
        if first > 0:        # the fp32 oracle on the same prefix: prompt + the bf16 oracle's first `first` ids
            ext32 = torch.cat([ext32, O.embed_tokens(sd, ref_ids[:, :first].long(), O.FP32)], 1)
        ref32_t = O.llm_forward(cfg, sd, ext32, P=O.FP32)[0][0, -1]
        band(f"generate A: step {first} logits under teacher forcing", hip_t, ref_t, ref32_t)
        assert int(hip_t.argmax()) == int(got[0, first]), "generate() and the teacher-forced step disagree on the same prefix"
        e_t = errs(hip_t, ref_t)[1]
        gap = float(ref_t[int(ref_ids[0, first])] - ref_t[int(got[0, first])])
        tol = 4.0 * e_t * float(ref_t.double().pow(2).mean().sqrt())
        record("generate A near-tie", step=first, oracle_gap=gap, tolerance=tol, step_rel_l2=e_t)
        assert gap < tol, f"step {first}: id {int(got[0, first])} vs oracle {int(ref_ids[0, first])}: the oracle's gap {gap:.3e} exceeds 4 x error x rms = {tol:.3e}"
    record("generate A", got=got[0].tolist(), oracle_bf16=ref_ids[0].tolist(), reference_fp32=g["A.generate.ids"][0].tolist(),
           min_margin=float(margin.min()))
    # against the reference's own fp32 greedy ids (golden): identical up to the first step whose fp32 top-2 margin
    # lies inside the measured bf16 logit band (random-weight logits have near-ties; after one the sequences
    # legitimately diverge).  The band (max-norm, ~1.8e-2 of max|logit| on this model) is measured in
    # test_forward_logits_case_a; 4e-2 = 2x that band + slack.
    emb32 = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=O.FP32)[4]
    ids32, lg32 = O.greedy_generate(cfg, sd, emb32, n_new, P=O.FP32, return_logits=True)
    assert ids32[0].tolist() == g["A.generate.ids"][0].tolist()
    t2 = lg32[0].topk(2, dim=-1).values
    m32 = (t2[:, 0] - t2[:, 1]) / lg32[0].abs().amax(-1)
    for t in range(n_new):
        if int(got[0, t]) != int(ids32[0, t]):
            assert float(m32[t]) < 4e-2, f"step {t}: differs from the fp32 reference although its margin is {float(m32[t]):.2e}"
            break
    assert got[0, 0].item() == int(g["A.generate.ids"][0, 0])


def test_generate_text_only_and_callbacks(env):
    cfg, model, g = env["cfg"], env["model"], env["gold"]
    ids = env["synth"].synth_ids(9, cfg.vocab, image_pos=-1, seed=7).unsqueeze(0)
    got = model.generate(input_ids=ids, images=None, do_sample=False, max_new_tokens=4, eos_token_id=None, pad_token_id=2)
    assert got[0].tolist() == g["C.generate.ids"][0].tolist()

    class Streamer:
        def __init__(self):
            self.puts, self.ended = [], False

        def put(self, v):
            self.puts.append(v.reshape(-1).tolist())

        def end(self):
            self.ended = True

    calls = []

    def stop_after_two(input_ids, scores, **kw):
        calls.append(tuple(input_ids.shape))
        return input_ids.shape[1] >= 2

    st = Streamer()
    got2 = model.generate(input_ids=ids, do_sample=False, max_new_tokens=4, eos_token_id=None, streamer=st,
                          stopping_criteria=[stop_after_two])
    assert got2[0].tolist() == g["C.generate.ids"][0].tolist()[:2]
    assert st.ended and st.puts[0] == [] and [p[0] for p in st.puts[1:]] == got2[0].tolist()
    assert calls == [(1, 1), (1, 2)]
    # eos stops generation and is included in the output
    first = g["C.generate.ids"][0].tolist()
    got3 = model.generate(input_ids=ids, do_sample=False, max_new_tokens=4, eos_token_id=[first[1]])
    assert got3[0].tolist() == first[:2]
    with pytest.raises(NotImplementedError):
        model.generate(input_ids=ids, inputs_embeds=torch.zeros(1, 2, 4096))


def test_decode_matches_prefill_logits(env):
    """teacher forcing: the logits of decode step t must equal the prefill logits at position S+t when the
    same tokens are fed (KV cache, RoPE positions and the GEMV path against the tile-GEMM path)."""
    cfg, model = env["cfg"], env["model"]
    eng = model.engine
    ids = env["synth"].synth_ids(40, cfg.vocab, image_pos=-1, seed=9)
    emb = eng.embed_tokens(ids).unsqueeze(0)
    full, _ = eng.prefill(emb, [40], all_logits=True, last_logits=False)
    S0 = 33
    _, last = eng.prefill(emb[:, :S0].contiguous(), [S0], all_logits=False, last_logits=True)
    check_close("prefill last-row path", last[0], full[0, S0 - 1], 2e-3, 1e-3)
    O, sd = env["O"], env["sd"]
    e_cpu = O.embed_tokens(sd, ids.unsqueeze(0), O.BF16)
    o16, _ = O.llm_forward(cfg, sd, e_cpu, P=O.BF16)
    o32, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    band("prefill text-only logits", full, o16, o32)
    for t in range(3):
        eng.decode_begin(ids[S0 + t:S0 + t + 1].to(torch.int32), [S0 + t], [S0 + t])
        _, lg = eng.decode_steps(1, return_logits=True)
        band(f"decode step {t} logits", lg[0, 0], o16[0, S0 + t], o32[0, S0 + t])


def test_rope_at_the_attention_q_load_is_bit_identical(env):
    """Round 3: the prefill attention rotates the queries as it loads them from the fused QKV row (rope_kv_kernel then only rotates K and
    appends K / V) - the same operation order as the kernel it replaces, so logits and the decode steps that follow are the same BITS as
    with the rotated copy of Q (vz_tune_set(33, 0)).  Ragged batch with left padding (positions are not row indices there)."""
    from vz_hip import binding as B
    cfg, model = env["cfg"], env["model"]
    eng = model.engine
    synth = env["synth"]
    for S in (40, 300):
        ids = synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=90 + S)
        emb = eng.embed_tokens(ids).unsqueeze(0)
        got = []
        for knob in (1, 0):
            try:
                B.check(B.lib().vz_tune_set(33, knob))
                full, last = eng.prefill(emb, [S], all_logits=True, last_logits=True)
                eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
                toks, lg = eng.decode_steps(3, return_logits=True)
                got.append((full.clone(), lg.clone(), toks.clone()))
            finally:
                B.check(B.lib().vz_tune_set(33, 1))
        assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1]) and torch.equal(got[0][2], got[1][2])
    if eng.max_batch >= 2:
        ids = torch.stack([synth.synth_ids(48, cfg.vocab, image_pos=-1, seed=s) for s in (5, 6)])
        mask = torch.ones_like(ids)
        mask[1, :17] = 0                                     # left padding on the second sample
        outs = []
        for knob in (1, 0):
            try:
                B.check(B.lib().vz_tune_set(33, knob))
                outs.append(model(input_ids=ids.to(model.device), attention_mask=mask.to(model.device)).logits.clone())
            finally:
                B.check(B.lib().vz_tune_set(33, 1))
        assert torch.equal(outs[0], outs[1])


def test_decode_graph_equals_eager(env):
    cfg, model = env["cfg"], env["model"]
    eng = model.engine
    ids = env["synth"].synth_ids(24, cfg.vocab, image_pos=-1, seed=11)
    emb = eng.embed_tokens(ids).unsqueeze(0)
    _, last = eng.prefill(emb, [24])
    first = last.argmax(-1).to(torch.int32)
    eng.decode_begin(first, [24], [24])
    eager, _ = eng.decode_steps(8, return_logits=True)       # debug logits force the eager path
    _, last = eng.prefill(emb, [24])
    eng.decode_begin(first, [24], [24])
    graph = eng.decode_steps(8)
    assert eager.tolist() == graph.tolist()


def test_batched_generate_and_left_padding(env):
    """true batching (one weight stream for all rows) must give every row exactly what it gets alone, and a left-padded
    batch (tokenizer_padding_side='left') must score like the same batch right-padded."""
    cfg, model = env["cfg"], env["model"]
    S = env["synth"]
    a = S.synth_ids(21, cfg.vocab, image_pos=-1, seed=21)
    b = S.synth_ids(14, cfg.vocab, image_pos=-1, seed=22)
    ids = torch.full((2, 21), 2, dtype=torch.long)
    ids[0] = a
    ids[1, :14] = b
    mask = torch.zeros(2, 21, dtype=torch.long)
    mask[0] = 1
    mask[1, :14] = 1
    both = model.generate(input_ids=ids, attention_mask=mask, do_sample=False, max_new_tokens=5, eos_token_id=None, pad_token_id=2)
    one_a = model.generate(input_ids=a.unsqueeze(0), do_sample=False, max_new_tokens=5, eos_token_id=None)
    one_b = model.generate(input_ids=b.unsqueeze(0), do_sample=False, max_new_tokens=5, eos_token_id=None)
    assert both.shape == (2, 5)
    assert both[0].tolist() == one_a[0].tolist() and both[1].tolist() == one_b[0].tolist()
    # eos: row 1 stops at its 2nd token and pads afterwards, row 0 continues
    stop = int(one_b[0, 1])
    if stop not in one_a[0].tolist():
        got = model.generate(input_ids=ids, attention_mask=mask, do_sample=False, max_new_tokens=5, eos_token_id=[stop], pad_token_id=0)
        assert got[1].tolist()[:2] == one_b[0].tolist()[:2] and set(got[1].tolist()[2:]) <= {0}
        assert got[0].tolist() == one_a[0].tolist()
    # every row hits eos: generation ends at the first step after which all rows are finished (row 0 at step 2, row 1 at step 0)
    e0, e1 = int(one_a[0, 2]), int(one_b[0, 0])
    if e0 not in one_a[0, :2].tolist() and e0 not in one_b[0].tolist() and e1 not in one_a[0, :3].tolist():
        got = model.generate(input_ids=ids, attention_mask=mask, do_sample=False, max_new_tokens=5, eos_token_id=[e0, e1], pad_token_id=0)
        assert got.shape == (2, 3)
        assert got[0].tolist() == one_a[0, :3].tolist() and got[1].tolist() == [e1, 0, 0]
    # left padding: roll row 1 to the right edge
    ids_l, mask_l = ids.clone(), mask.clone()
    ids_l[1] = torch.roll(ids[1], 7)
    mask_l[1] = torch.roll(mask[1], 7)
    lr = model(input_ids=ids, attention_mask=mask).logits
    ll = model(input_ids=ids_l, attention_mask=mask_l).logits
    assert torch.equal(ll[0], lr[0])
    assert torch.equal(ll[1, 7:], lr[1, :14])
    gen_l = model.generate(input_ids=ids_l, attention_mask=mask_l, do_sample=False, max_new_tokens=5, eos_token_id=None, pad_token_id=2)
    assert gen_l.tolist() == both.tolist()


def test_sampling_is_seeded_and_follows_the_logits(env):
    cfg, model = env["cfg"], env["model"]
    ids = env["synth"].synth_ids(9, cfg.vocab, image_pos=-1, seed=7).unsqueeze(0)
    g1 = torch.Generator(device=model.device).manual_seed(123)
    g2 = torch.Generator(device=model.device).manual_seed(123)
    s1 = model.generate(input_ids=ids, do_sample=True, temperature=0.7, top_p=0.9, max_new_tokens=6, eos_token_id=None, generator=g1)
    s2 = model.generate(input_ids=ids, do_sample=True, temperature=0.7, top_p=0.9, max_new_tokens=6, eos_token_id=None, generator=g2)
    assert s1.tolist() == s2.tolist() and s1.shape == (1, 6)
    # temperature -> 0 collapses onto the greedy path
    greedy = model.generate(input_ids=ids, do_sample=False, max_new_tokens=4, eos_token_id=None)
    cold = model.generate(input_ids=ids, do_sample=True, temperature=1e-4, max_new_tokens=4, eos_token_id=None)
    assert cold.tolist() == greedy.tolist()


def test_cls_patch_feature_selection(env):
    """mm_vision_select_feature = 'cls_patch' (ref vision_encoder.py:66-73): the class token stays in, 577 visual tokens per
    tile flow through the fusion and into the Q-Former's cross-attention.  Golden case D + the bf16 band."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    O, sd, synth = env["O"], env["sd"], env["synth"]
    cfg = synth.ArchConfig(n_layers=1, clip_keep_cls=True)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=1,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    hf.mm_vision_select_feature = "cls_patch"
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=2, max_text=64)
    assert model.arch.clip_keep_cls and model.arch.vision_tokens == 577
    tiles = synth.synth_tiles(2, seed=8)
    ids = synth.synth_ids(32, cfg.vocab, image_pos=5, seed=2)
    te = O.embed_tokens(sd, ids[ids != -200], O.BF16).unsqueeze(0).expand(2, -1, -1)
    fused = model.get_vision_tower()(tiles.to(model.device).bfloat16())
    assert tuple(fused.shape) == (2, 577, 5120)
    band("fused cls_patch", fused, O.clip_tower(cfg, sd, tiles, O.BF16), O.clip_tower(cfg, sd, tiles, O.FP32))
    _gold_check(env, "D.fused.cls_patch", fused, 5e-2, 1.5e-2)
    feats = O.clip_tower(cfg, sd, tiles, O.BF16)                      # same features into both sides
    out = model.engine.qformer(feats.bfloat16(), te.bfloat16()[:1].contiguous(), [0, 0])
    band("qformer cls_patch", out, O.qformer(cfg, sd, feats, te, O.BF16), O.qformer(cfg, sd, feats, te, O.FP32))
    enc = model.encode_images(tiles.to(model.device).bfloat16(), te.bfloat16().to(model.device))
    _gold_check(env, "D.encode_images.cls_patch", enc, 1e-1, 3e-2)
    del model
    torch.cuda.empty_cache()


def test_resize_token_embeddings_live(env):
    """ref builder.py:141-153 adds <im_patch> and calls model.resize_token_embeddings(32001): the live engine grows its
    embedding / lm_head tables (new row = mean of the old rows), old logits stay bit-identical, the new id embeds to the mean."""
    model, cfg = env["model"], env["cfg"]
    eng = model.engine
    ids = env["synth"].synth_ids(24, cfg.vocab, image_pos=-1, seed=11).unsqueeze(0).to(model.device)
    before = model(input_ids=ids).logits
    mean_row = eng.w["llm.embed"].float().mean(0).bfloat16()
    try:
        model.resize_token_embeddings(cfg.vocab + 1)
        assert model.config.vocab_size == cfg.vocab + 1 and eng.cfg.vocab == cfg.vocab + 1
        after = model(input_ids=ids).logits
        assert after.shape[-1] == cfg.vocab + 1
        assert torch.equal(after[..., :cfg.vocab], before)
        emb = model.get_model().embed_tokens(torch.tensor([cfg.vocab], device=model.device))
        assert torch.equal(emb[0], mean_row)
        out = model.generate(input_ids=ids, do_sample=False, max_new_tokens=4, eos_token_id=None, pad_token_id=2)
        assert out.shape == (1, 4) and int(out.max()) <= cfg.vocab
    finally:
        model.resize_token_embeddings(cfg.vocab)
    assert torch.equal(model(input_ids=ids).logits, before)


def test_nf4_weight_engine_matches_quantized_oracle(env):
    """`load_4bit` of the drop-in builder (ref:vis_zephyr/model/builder.py:35-43): the decoder layers' linears NF4-quantised at load time
    (64-element blocks, vz_hip/quant.py), every kernel computing on the dequantised bf16 values.  Against the oracle running the
    identically quantised state dict: engine tensors bit-equal, prefill logits inside the bf16 band, the quantisation visible against the
    unquantised oracle, decode = prefill under teacher forcing.  (Parity against bitsandbytes itself: unpinned, see vz_hip/quant.py.)"""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    O, sd, synth, cfg = env["O"], env["sd"], env["synth"], env["cfg"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=1, max_text=64, weight_nf4=True)
    eng = model.engine
    sdq = O.quantize_state_dict_nf4(sd)
    qkv = torch.cat([sdq[f"model.layers.1.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    assert torch.equal(eng.w["llm.1.qkv.w"].float().cpu(), qkv)
    g = eng.w["llm.0.gu.w"].float().cpu().view(-1, 2, 16, cfg.hidden)
    assert torch.equal(g[:, 0].reshape(-1, cfg.hidden), sdq["model.layers.0.mlp.gate_proj.weight"])
    assert torch.equal(eng.w["llm.lm_head"].float().cpu(), sd["lm_head.weight"].bfloat16().float())          # not quantised (skip list)
    # 16 levels x one scale per block: at most 16 distinct |values| / absmax in a block
    blk = eng.w["llm.0.down.w"][5, :64].float()
    assert len(torch.unique((blk / blk.abs().max()).round(decimals=3))) <= 16
    ids = synth.synth_ids(40, cfg.vocab, image_pos=-1, seed=9)
    lo_bf, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.BF16), P=O.BF16)
    lo_32, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    emb = eng.embed_tokens(ids).unsqueeze(0)
    full, _ = eng.prefill(emb, [40], all_logits=True, last_logits=False)
    band("nf4 prefill logits", full, lo_bf, lo_32)
    lo_unq, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    assert errs(full, lo_unq)[1] > 2 * errs(full, lo_32)[1]
    S0 = 33
    eng.prefill(emb[:, :S0].contiguous(), [S0], all_logits=False, last_logits=True)
    eng.decode_begin(ids[S0:S0 + 1].to(torch.int32), [S0], [S0])
    _, lg = eng.decode_steps(1, return_logits=True)
    band("nf4 decode step logits", lg[0, 0], lo_bf[0, S0], lo_32[0, S0])
    # a second finalize (e.g. after load_state_dict of new projector weights) does not quantise twice
    before = eng.w["llm.1.qkv.w"].clone()
    eng.finalize()
    assert torch.equal(eng.w["llm.1.qkv.w"], before)
    del model
    torch.cuda.empty_cache()


def test_fp8_weight_engine_matches_quantized_oracle(env):
    """W8A16 (SURVEY config 5; `load_8bit` of the drop-in builder): the engine's decode GEMVs stream e4m3 weights with one
    power-of-two scale per row, its prefill GEMMs run on the bf16 copy of the same dequantised weights.  Checked against
    the oracle running the identically quantised state dict: packed weights bit-exact, prefill logits in the bf16 band,
    decode (fp8 stream) = prefill (bf16 MFMA) under teacher forcing, greedy ids up to oracle near-ties."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    from vz_hip import quant
    O, sd, synth, cfg = env["O"], env["sd"], env["synth"], env["cfg"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=1, max_text=64, weight_fp8=True)
    eng = model.engine
    sdq = O.quantize_state_dict(sd)
    # the engine's bf16 tensors hold exactly the dequantised e4m3 weights, and those are the oracle's
    qkv = torch.cat([sdq[f"model.layers.1.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    assert torch.equal(eng.w["llm.1.qkv.w"].float().cpu(), qkv)
    assert torch.equal(quant.dequantize_rows(eng.w["llm.1.qkv.w8"], eng.w["llm.1.qkv.ws"]).cpu(), qkv)
    assert torch.equal(eng.w["llm.lm_head"].float().cpu(), sdq["lm_head.weight"])
    g = eng.w["llm.0.gu.w"].float().cpu().view(-1, 2, 16, cfg.hidden)
    assert torch.equal(g[:, 0].reshape(-1, cfg.hidden), sdq["model.layers.0.mlp.gate_proj.weight"])
    assert torch.equal(eng.w["llm.embed"].float().cpu(), sd["model.embed_tokens.weight"].bfloat16().float())   # not quantised
    # prefill logits (bf16 MFMA on dequantised weights) in the bf16 band of the quantised-weight oracle
    ids = synth.synth_ids(40, cfg.vocab, image_pos=-1, seed=9)
    x_or = O.embed_tokens(sd, ids.unsqueeze(0), O.BF16)
    lo_bf, _ = O.llm_forward(cfg, sdq, x_or, P=O.BF16)
    lo_32, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    emb = eng.embed_tokens(ids).unsqueeze(0)
    full, _ = eng.prefill(emb, [40], all_logits=True, last_logits=False)
    band("fp8 prefill logits", full, lo_bf, lo_32)
    # quantisation is visible: the unquantised oracle is measurably further away than the band
    lo_unq, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    assert errs(full, lo_unq)[1] > 2 * errs(full, lo_32)[1]
    # teacher forcing: decode steps (fp8 weight stream) reproduce the prefill logits (bf16 MFMA GEMMs)
    S0 = 33
    _, last = eng.prefill(emb[:, :S0].contiguous(), [S0], all_logits=False, last_logits=True)
    check_close("fp8 last-row logits (GEMV) vs all-rows (GEMM)", last[0], full[0, S0 - 1], 3e-2, 6e-3)
    eng.decode_begin(ids[S0:S0 + 1].to(torch.int32), [S0], [S0])
    _, lg = eng.decode_steps(1, return_logits=True)
    # the decode step (GEMV/skinny weight stream, split softmax) and the prefill row (MFMA GEMMs, flash
    # attention) are two bf16 evaluation orders of the same network: they sit inside the same band the
    # oracle's own bf16-vs-fp32 distance defines, not inside a hand-picked constant
    e_or = errs(lo_bf, lo_32)[1]
    check_close("fp8 decode step vs prefill", lg[0, 0], full[0, S0], 3e-2, band_tol(e_or))
    # greedy generation against the quantised-weight oracle
    n_new = 6
    got = model.generate(input_ids=ids.unsqueeze(0), do_sample=False, max_new_tokens=n_new, eos_token_id=None, pad_token_id=2)
    ref_ids, ref_logits = O.greedy_generate(cfg, sdq, x_or, n_new, P=O.BF16, return_logits=True)
    top2 = ref_logits[0].topk(2, dim=-1).values
    margin = (top2[:, 0] - top2[:, 1]) / ref_logits[0].abs().amax(-1)
    for t in range(n_new):
        if int(got[0, t]) != int(ref_ids[0, t]):
            assert float(margin[t]) < 2e-3, f"step {t}: id {int(got[0, t])} vs oracle {int(ref_ids[0, t])}, margin {float(margin[t]):.2e}"
            break
    record("generate fp8", got=got[0].tolist(), oracle_bf16_quantized=ref_ids[0].tolist(), min_margin=float(margin.min()))
    # A SECOND load into the same fp8 engine (base then finetune, LoRA re-merge): the e4m3 copies are derived data and must follow
    # the rewritten bf16 tensors - decode (fp8 stream) and prefill (bf16) keep computing with ONE model.
    sd2 = {k: v.cpu() for k, v in synth.iter_state_dict(cfg, 1, device=model.device, prefixes=("model.layers.", "lm_head", "model.norm", "model.embed"))}
    eng.load_weights(sd2.items())
    sdq2 = O.quantize_state_dict({**sd, **sd2})
    qkv2 = torch.cat([sdq2[f"model.layers.1.self_attn.{n}_proj.weight"] for n in "qkv"], 0)
    assert not torch.equal(qkv2, qkv)
    assert torch.equal(eng.w["llm.1.qkv.w"].float().cpu(), qkv2)
    assert torch.equal(quant.dequantize_rows(eng.w["llm.1.qkv.w8"], eng.w["llm.1.qkv.ws"]).cpu(), qkv2)
    emb2 = eng.embed_tokens(ids).unsqueeze(0)
    full2, _ = eng.prefill(emb2, [40], all_logits=True, last_logits=False)
    eng.prefill(emb2[:, :S0].contiguous(), [S0], all_logits=False, last_logits=True)
    eng.decode_begin(ids[S0:S0 + 1].to(torch.int32), [S0], [S0])
    _, lg2 = eng.decode_steps(1, return_logits=True)
    check_close("fp8 decode step vs prefill after a second load", lg2[0, 0], full2[0, S0], 3e-2, band_tol(e_or))
    # resize_token_embeddings on the fp8 engine: lm_head8 / lm_heads are rebuilt at the new size (finalize used to fail on the stale ones)
    model.resize_token_embeddings(cfg.vocab + 1)
    assert eng.w["llm.lm_head8"].shape[0] == cfg.vocab + 1 and eng.w["llm.lm_heads"].shape[0] == cfg.vocab + 1
    out_r = model.generate(input_ids=ids.unsqueeze(0), do_sample=False, max_new_tokens=3, eos_token_id=None, pad_token_id=2)
    assert out_r.shape == (1, 3)
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("Bn", [5, 16, 19, 32, 41, 64])
def test_wide_batch_decode_matches_prefill(env, Bn):
    """5..64 rows decode through the MFMA weight stream (gemm_skinny.hip; 17..64 rows: two / four B operands per weight fragment, the
    RMSNorm once into an L2-resident scratch):
    under teacher forcing every row's decode-step logits must equal its prefill logits at the same position (KV slots,
    per-row positions / lengths, ragged prompts), and batched greedy generation must give each row what it gets alone."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg, synth = env["cfg"], env["synth"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=64, max_ctx=128, max_tiles=1, max_text=16)
    eng = model.engine
    lens = [20 + (7 * b) % 13 for b in range(Bn)]
    Smax = max(lens) + 1
    ids = torch.full((Bn, Smax), 2, dtype=torch.long)
    for b in range(Bn):
        ids[b, : lens[b] + 1] = synth.synth_ids(lens[b] + 1, cfg.vocab, image_pos=-1, seed=40 + b)
    emb = eng.embed_tokens(ids)                                       # [B, Smax, H]
    full, _ = eng.prefill(emb, [l + 1 for l in lens], all_logits=True, last_logits=False)      # logits at every position
    eng.prefill(emb, lens, all_logits=False, last_logits=True)       # cache holds positions < len; the next token is ids[b, len]
    eng.decode_begin(torch.tensor([int(ids[b, lens[b]]) for b in range(Bn)], dtype=torch.int32), lens, lens)
    _, lg = eng.decode_steps(1, return_logits=True)
    for b in range(Bn):
        check_close(f"B{Bn} row {b} decode step vs prefill", lg[0, b], full[b, lens[b]], 3e-2, 1e-2)      # two bf16 paths through 2 layers: ~6e-3
    # generation: rows of the batch against the same prompts alone (greedy; a near-tie may flip a token: compare up to it)
    mask = torch.zeros(Bn, Smax, dtype=torch.long)
    for b in range(Bn):
        mask[b, : lens[b]] = 1
    both = model.generate(input_ids=ids, attention_mask=mask, do_sample=False, max_new_tokens=4, eos_token_id=None, pad_token_id=2)
    assert both.shape == (Bn, 4)
    same = 0
    for b in (0, Bn // 2, Bn - 1):
        one = model.generate(input_ids=ids[b:b + 1, : lens[b]], do_sample=False, max_new_tokens=4, eos_token_id=None)
        assert int(both[b, 0]) == int(one[0, 0])                     # the first token comes from the same prefill kernels
        same += int(both[b].tolist() == one[0].tolist())
    assert same >= 2
    del model
    torch.cuda.empty_cache()


def test_sliding_window_and_context_capacity(env):
    """Mistral's sliding window (hf:models/mistral/modeling_mistral.py:364-371; Zephyr: 4096) at engine level with a small window:
    prefill of 100 tokens and teacher-forced decode steps against the oracle's windowed mask, then the capacity edge:
    prompt + new tokens == max_ctx runs, one more raises."""
    import dataclasses
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    O, sd, synth = env["O"], env["sd"], env["synth"]
    cfg = dataclasses.replace(env["cfg"], sliding_window=48)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=48,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=112, max_tiles=1, max_text=16)
    eng = model.engine
    S = 100
    ids = synth.synth_ids(S + 4, cfg.vocab, image_pos=-1, seed=31)
    o16, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids.unsqueeze(0), O.BF16), P=O.BF16)
    o32, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    # the window matters: the same tokens under the full causal mask give visibly different logits late in the sequence
    full32, _ = O.llm_forward(env["cfg"], sd, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    assert errs(o32[0, 90], full32[0, 90])[1] > 0.05
    emb = eng.embed_tokens(ids).unsqueeze(0)
    got, _ = eng.prefill(emb[:, :S].contiguous(), [S], all_logits=True, last_logits=False)
    band("windowed prefill logits", got, o16[:, :S], o32[:, :S])
    for t in range(3):                                              # decode steps S, S+1, S+2 look back 48 positions only
        eng.decode_begin(ids[S + t:S + t + 1].to(torch.int32), [S + t], [S + t])
        _, lg = eng.decode_steps(1, return_logits=True)
        band(f"windowed decode step {t}", lg[0, 0], o16[0, S + t], o32[0, S + t])
    # capacity: 100 + 12 == max_ctx is served, 100 + 13 is refused before anything is launched
    out = model.generate(input_ids=ids[:S].unsqueeze(0), do_sample=False, max_new_tokens=12, eos_token_id=None)
    assert out.shape == (1, 12)
    with pytest.raises(ValueError):
        model.generate(input_ids=ids[:S].unsqueeze(0), do_sample=False, max_new_tokens=13, eos_token_id=None)
    # the same edge at the C ABI (a caller that steps the engine itself): the n steps of a call must fit the cache row and the
    # rotary tables, checked on the host before anything is launched ...
    eng.decode_begin(ids[:1].to(torch.int32), [110], [110])          # 110 cached, the next token sees 111 keys
    with pytest.raises(ValueError, match="max_ctx"):
        eng.decode_steps(3)                                            # 111 + 3 - 1 = 113 > 112
    assert eng.decode_steps(2).shape == (1, 2)                          # 112: the last slot of the row
    with pytest.raises(ValueError, match="max_ctx"):
        eng.decode_steps(1)
    # ... while a PARKED row (continuous batching: context 0, position 0) may step for ever: the step tail saturates its slot /
    # length / position on the device instead of walking out of its cache row
    eng.decode_begin(torch.zeros(1, dtype=torch.int32), [0], [0])
    for _ in range(3):
        eng.decode_steps(64)
    torch.cuda.synchronize()
    out2 = model.generate(input_ids=ids[:S].unsqueeze(0), do_sample=False, max_new_tokens=12, eos_token_id=None)
    assert out2.tolist() == out.tolist()
    del model
    torch.cuda.empty_cache()


def test_full_size_request_properties(env):
    """BASELINE.json configs[2] at full size (5 anyres tiles + 1889 ids -> S = 2048; 2 decoder layers keep it quick), checked
    through size-independent properties instead of the CPU oracle (which needs minutes there):
      * tiles are independent units: encoding a permuted tile list permutes the visual tokens (bit for bit with whole-tile GEMMs);
      * generate's last-row logits path and forward's all-position path agree on the first token;
      * teacher forcing at context 2048: the decode step's logits equal the prefill logits of the 2049-token sequence;
      * the captured decode graph and the eager step produce the same ids."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg, synth = env["cfg"], env["synth"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=2048 + 64, max_tiles=5, max_text=2048)
    eng = model.engine
    tiles = synth.synth_tiles(5, seed=1).to(model.device).bfloat16()
    ids = synth.synth_ids(1889, cfg.vocab, image_pos=5, seed=2).unsqueeze(0).to(model.device)
    # tile independence
    perm = torch.tensor([3, 0, 4, 1, 2], device=model.device)
    text = eng.embed_tokens(ids[0][ids[0] != -200]).unsqueeze(0)
    from vz_hip import binding as B
    a = model.encode_images(tiles, text, tile_sample=[0] * 5)
    b = model.encode_images(tiles[perm], text, tile_sample=[0] * 5)
    assert a.shape == (5, 32, cfg.hidden)
    # production dispatch: the stream-K tail of the 256x256 GEMM sums K in slices for the rows that fall into the split tiles
    # (fp32 re-association), so a moved tile agrees to bf16 rounding through the 8 Q-Former blocks ...
    check_close("permuted tiles (production dispatch)", b, a[perm].float(), 0.25, 2e-2)
    try:                                                              # ... and bit for bit with whole tiles only
        B.check(B.lib().vz_tune_set(4, 0))
        a0 = model.encode_images(tiles, text, tile_sample=[0] * 5)
        b0 = model.encode_images(tiles[perm], text, tile_sample=[0] * 5)
    finally:
        B.check(B.lib().vz_tune_set(4, 1))
    assert torch.equal(b0, a0[perm])
    # the Q-Former at the full text length (block 0 attends over 32 + 1888 rows) for one tile, against the oracle's band
    O, sd = env["O"], env["sd"]
    feats1 = O.clip_tower(cfg, sd, synth.synth_tiles(5, seed=1)[:1], O.BF16)
    te = O.embed_tokens(sd, ids[0][ids[0] != -200].cpu(), O.BF16).unsqueeze(0)
    qf_hip = eng.qformer(feats1.bfloat16(), te.bfloat16(), [0])
    band("qformer at L=1888", qf_hip, O.qformer(cfg, sd, feats1, te, O.BF16), O.qformer(cfg, sd, feats1, te, O.FP32))
    # forward (all positions) vs generate (last row)
    logits = model(input_ids=ids, images=[tiles]).logits
    assert logits.shape == (1, 2048, cfg.vocab)
    out = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=9, eos_token_id=None)
    assert out.shape == (1, 9) and int(out[0, 0]) == int(logits[0, -1].argmax())
    # teacher forcing at full context: prefill of S+1 embeddings vs prefill of S + one decode step
    emb = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles])[4]
    nxt = out[0, :1].to(torch.int32)
    emb1 = torch.cat([emb, eng.embed_tokens(nxt.to(torch.long)).view(1, 1, -1)], dim=1)
    full, _ = eng.prefill(emb1, [2049], all_logits=True, last_logits=False)
    eng.prefill(emb, [2048], all_logits=False, last_logits=True)
    eng.decode_begin(nxt, [2048], [2048])
    ids_e, lg = eng.decode_steps(4, return_logits=True)              # debug logits: eager steps
    check_close("decode step at ctx 2048 vs prefill of 2049", lg[0, 0], full[0, 2048], 3e-2, 1e-2)
    eng.prefill(emb, [2048], all_logits=False, last_logits=True)
    eng.decode_begin(nxt, [2048], [2048])
    ids_g = eng.decode_steps(4)                                       # captured graph
    assert torch.equal(ids_g, ids_e) and ids_g[0].tolist() == out[0, 1:5].tolist()
    del model
    torch.cuda.empty_cache()


def test_medium_prompt_batch_against_oracle(env):
    """several query blocks per head and both tile GEMMs in play (the S = 127 cases above fit one attention block and one GEMM
    row tile): a ragged batch of 2 text prompts, 450 and 333 tokens, prefill logits of every valid position in the oracle's band."""
    O, cfg, sd = env["O"], env["cfg"], env["sd"]
    eng = env["model"].engine
    lens = [450, 333]                      # env engine: max_ctx 512
    Smax = max(lens)
    ids = torch.full((2, Smax), 2, dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, :n] = env["synth"].synth_ids(n, cfg.vocab, image_pos=-1, seed=50 + b)
    emb = eng.embed_tokens(ids)
    full, _ = eng.prefill(emb, lens, all_logits=True, last_logits=False)
    for b, n in enumerate(lens):
        x16 = O.embed_tokens(sd, ids[b:b + 1, :n], O.BF16)
        x32 = O.embed_tokens(sd, ids[b:b + 1, :n], O.FP32)
        o16, _ = O.llm_forward(cfg, sd, x16, P=O.BF16)
        o32, _ = O.llm_forward(cfg, sd, x32, P=O.FP32)
        band(f"medium prompt row {b} ({n} tokens)", full[b, :n], o16[0], o32[0])


def test_continuous_batching_matches_static_batches(env):
    """generate_stream (continuous batching): 7 requests of different prompt lengths and token budgets through 3 KV-cache
    rows, one of them multimodal, some ending on an eos token mid-chunk; every sequence must be exactly what the static
    batched path produces for it (rows are independent and the 2..16-row kernels are batch-invariant), whatever the
    order of admission and completion."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg, synth = env["cfg"], env["synth"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=6, max_ctx=256, max_tiles=4, max_text=64)
    lens = [9, 30, 17, 44, 12, 25, 38]
    budgets = [5, 21, 9, 33, 1, 18, 7]
    tiles = synth.synth_tiles(2, seed=3).to(model.device).bfloat16()
    reqs = []
    for i, (n, bud) in enumerate(zip(lens, budgets)):
        ids = synth.synth_ids(n, cfg.vocab, image_pos=4 if i == 3 else -1, seed=70 + i).unsqueeze(0)
        r = {"input_ids": ids, "max_new_tokens": bud}
        if i == 3:
            r["images"] = tiles
        reqs.append(r)

    def static(i, eos):     # the same request as row 0 of a 2-row static batch (same kernels as the stream's rows)
        ids = reqs[i]["input_ids"].to(model.device)
        two = torch.cat([ids, ids], 0)
        kw = dict(images=[tiles, tiles]) if "images" in reqs[i] else {}
        out = model.generate(input_ids=two, do_sample=False, max_new_tokens=budgets[i], eos_token_id=eos, pad_token_id=0, **kw)
        assert torch.equal(out[0], out[1])
        return out[0].tolist()

    # The multimodal request runs with 4 tiles in the static batch and 2 in the stream: under the production dispatch the 256^2
    # GEMM's stream-K tail sums K in slices for the rows that fall into its split tiles, so a tile's visual tokens depend (in the last
    # bf16 bit) on its position in the batch - ids would then agree only up to near-ties.  Whole tiles only (knob 4 = 0) is the
    # batch-invariant kernel choice (as in test_full_size_request_properties): identical samples give identical rows, bit for bit.
    from vz_hip import binding as B
    try:
        B.check(B.lib().vz_tune_set(4, 0))
        B.check(B.lib().vz_tune_set(26, 0))        # and no tile-count-dependent K slices in the CLIP tower's small grids
        free_run = {i: static(i, None) for i in range(len(reqs))}
        # an eos that request 1 emits at its 8th token (mid-chunk with sync_every = 4) and that hits nobody's first token
        eos = free_run[1][7]
        assume_ok = all(f[0] != eos for f in free_run.values())
        # admit = 0: every request is prefilled alone, straight into its row; admit = None: the requests entering at one sync share
        # one right-padded prefill in the 3 spare cache rows and their KV is moved to the freed rows (vz_llm_kv_move_rows);
        # admit = 2: groups of two, so a sync with three free rows takes a batched and a single admission
        for admit in (0, None, 2):
            got = dict(model.generate_stream(reqs, eos_token_id=[eos] if assume_ok else None, rows=3, sync_every=4, admit=admit))
            assert sorted(got) == list(range(len(reqs)))
            for i in range(len(reqs)):
                want = free_run[i]
                if assume_ok and eos in want:
                    want = want[: want.index(eos) + 1]
                assert got[i].tolist() == want, f"admit={admit} request {i}: {got[i].tolist()} vs {want}"
        with pytest.raises(ValueError):
            next(model.generate_stream(reqs, rows=3, admit=4))            # only 3 spare rows
        with pytest.raises(ValueError):
            model.engine.kv_move_rows([4, 5], [0, 4], [8, 8])            # row 4 read by one move and written by the other
        # one row only degenerates to sequential generation and still agrees
        got1 = dict(model.generate_stream(reqs[:3], eos_token_id=None, rows=1, sync_every=16))
        for i in range(3):
            g1, w = got1[i].tolist(), free_run[i]
            assert g1[0] == w[0] and len(g1) == len(w)           # (a 1-row step runs the GEMV: a near-tie may flip later tokens)
    finally:
        B.check(B.lib().vz_tune_set(4, 1))
        B.check(B.lib().vz_tune_set(26, 1))
    env["_cb"] = dict(model=model, reqs=reqs, free_run=free_run, tiles=tiles, budgets=budgets)


def test_continuous_batching_default_knobs_near_tie(env):
    """The same stream at the SHIPPED knobs (stream-K tail and tile-count-dependent K slices on): a shared admission prefill / another
    row count may take another - equally accurate - bf16 evaluation order than the lone request, so the ids are held to: first token
    equal, and where a sequence leaves the pinned-route ids the two candidates are a near-tie in the request's own logits (gap below a
    few bf16-band errors of that row).  Documents the tolerance `generate_stream`'s docstring states."""
    cb = env.pop("_cb", None)
    if cb is None:
        pytest.skip("needs the engine of test_continuous_batching_matches_static_batches")
    model, reqs, free_run, tiles, budgets = cb["model"], cb["reqs"], cb["free_run"], cb["tiles"], cb["budgets"]
    try:
        got = dict(model.generate_stream(reqs, eos_token_id=None, rows=3, sync_every=4, admit=None))
        flips = []
        for i in range(len(reqs)):
            g, w = got[i].tolist(), free_run[i]
            assert len(g) == len(w) == budgets[i] and g[0] == w[0], (i, g, w)
            first = next((t for t in range(len(w)) if g[t] != w[t]), -1)
            if first < 0:
                continue
            ids = torch.cat([reqs[i]["input_ids"][0], torch.tensor(w[:first], dtype=torch.long)]).unsqueeze(0)
            kw = dict(images=[tiles]) if "images" in reqs[i] else {}
            row = model(input_ids=ids, **kw).logits[0, -1].double().cpu()
            gap = abs(float(row[g[first]] - row[w[first]]))
            tol = 4.0 * 2e-2 * float(row.pow(2).mean().sqrt())          # 2e-2: the bf16 band of this 2-layer model's logits (band("logits A"))
            flips.append(dict(request=i, step=first, gap=gap, tolerance=tol))
            assert gap < tol, f"request {i}: the stream leaves the lone request's ids at step {first} with a logit gap {gap:.3e} (tolerance {tol:.3e})"
        record("continuous batching at default knobs", flips=flips)
    finally:
        del model, cb
        torch.cuda.empty_cache()


def test_stage1_shaped_batch_forward(env):
    """BASELINE.json configs[3] (Stage-1 pretrain shape) without the tensor parallelism: one `forward` over a batch of 24 plain-
    caption samples with 3..5 anyres tiles each (97 tiles, more than the engine holds at once -> tile chunks; more samples than
    KV rows -> prefill chunks), labels in, loss out.  Rows must not depend on the batch around them: three of them are compared
    with their single-sample forward, and the loss with the one recomputed from those logits."""
    cfg, model, synth = env["cfg"], env["model"], env["synth"]          # env engine: max_batch 2, max_tiles 4, max_ctx 512
    Bn = 24
    n_tiles = [3 + (b * 7) % 3 for b in range(Bn)]
    lens = [12 + (b * 5) % 17 for b in range(Bn)]
    L = max(lens)
    ids = torch.full((Bn, L), 2, dtype=torch.long)
    mask = torch.zeros(Bn, L, dtype=torch.long)
    labels = torch.full((Bn, L), -100, dtype=torch.long)
    tiles = []
    for b in range(Bn):
        row = synth.synth_ids(lens[b], cfg.vocab, image_pos=1, seed=200 + b)        # plain caption: image right after BOS
        ids[b, :lens[b]] = row
        mask[b, :lens[b]] = 1
        labels[b, 2:lens[b]] = row[2:]
        tiles.append(synth.synth_tiles(n_tiles[b], seed=300 + b).to(model.device).bfloat16())
    out = model(input_ids=ids.to(model.device), attention_mask=mask.to(model.device), labels=labels.to(model.device), images=tiles)
    Smax = max(lens[b] - 1 + 32 * n_tiles[b] for b in range(Bn))
    assert out.logits.shape == (Bn, Smax, cfg.vocab) and bool(torch.isfinite(out.loss))
    for b in (0, 11, 23):
        # alone, but with the SAME padded ids: the reference conditions the Q-Former on input_ids[i] pads included and without a
        # key-padding mask, so a sample's visual tokens depend on the padded length of its batch
        one = model(input_ids=ids[b:b + 1].to(model.device), attention_mask=mask[b:b + 1].to(model.device), images=[tiles[b]]).logits[0]
        S = lens[b] - 1 + 32 * n_tiles[b]
        # two valid bf16 paths (the tile GEMMs see different row counts -> different split-K / stream-K sums) decorrelate to the bf16
        # band of the logits (~1.5e-2, test_forward_logits_case_a); a wrong sample map or cache row gives O(1)
        check_close(f"stage-1 batch row {b} vs alone", out.logits[b, :S], one[:S], 6e-2, 2.5e-2)
    # the loss is the mean token cross-entropy over the caption positions of the spliced sequences
    _, _, _, _, _, lab = model.prepare_inputs_labels_for_multimodal(ids.to(model.device), None, mask.to(model.device), None,
                                                                    labels.to(model.device), tiles)
    ref = torch.nn.functional.cross_entropy(out.logits[:, :-1].reshape(-1, cfg.vocab), lab[:, 1:].reshape(-1), ignore_index=-100)
    assert abs(float(out.loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert int((lab != -100).sum()) == sum(l - 2 for l in lens)


def test_forward_loss_on_device_and_cache_continuation(env):
    """a2: the loss of forward(labels=...) is computed by the HIP cross-entropy (hf ForCausalLMLoss: shift, ignore -100, mean over the valid
    targets) and equals torch's on the returned logits; forward(new ids, past_key_values=<handle>) continues on the engine's cache - its
    logits equal the full forward's at those positions (two bf16 evaluation orders), a stale handle is refused."""
    cfg, model, synth = env["cfg"], env["model"], env["synth"]
    n = 37
    ids = synth.synth_ids(n + 3, cfg.vocab, image_pos=-1, seed=77).unsqueeze(0).cuda()
    labels = ids.clone()
    labels[0, :5] = -100
    labels[0, 20] = -100
    full = model(input_ids=ids, labels=labels)
    ref = torch.nn.functional.cross_entropy(full.logits[:, :-1].reshape(-1, cfg.vocab).float(), labels[:, 1:].reshape(-1), ignore_index=-100)
    assert abs(float(full.loss) - float(ref)) <= 2e-5 * abs(float(ref)), (float(full.loss), float(ref))
    none = model(input_ids=ids, labels=torch.full_like(ids, -100))
    assert torch.isnan(none.loss)
    head = model(input_ids=ids[:, :n])
    stale = head.past_key_values
    step = model(input_ids=ids[:, n:n + 2], past_key_values=head.past_key_values)
    assert step.logits.shape == (1, 2, cfg.vocab) and step.past_key_values.lengths == [n + 2]
    for t in range(2):
        check_close(f"continued forward, new token {t}", step.logits[0, t], full.logits[0, n + t], 3e-2, 1e-2)
    last = model(input_ids=ids[:, n + 2:n + 3], past_key_values=step.past_key_values)
    check_close("continued forward, third token", last.logits[0, 0], full.logits[0, n + 2], 3e-2, 1e-2)
    with pytest.raises(ValueError):
        model(input_ids=ids[:, n:n + 1], past_key_values=stale)


def test_fp8_mfma_prefill_matches_quantized_oracle(env):
    """Config 5's "fp8 MFMA weights": on a weight_fp8 engine with set_prefill_fp8() the Zephyr prefill linears run e4m3 x e4m3 on the scaled
    MFMA - inputs quantised per row on the device, weights = the e4m3 copies.  Against the oracle that quantises IDENTICALLY (quantised
    state dict + the same per-row activation quantiser in front of every prefill linear): logits inside the band the oracle's own
    bf16-vs-fp32 distance defines (roundings may flip an e4m3 step, which that distance contains too); the cost of e4m3 activations against
    the weight-only-quantised oracle is recorded and bounded by the oracle's own; decode steps (W8A16 stream, bf16 activations) continue from the fp8-prefilled
    cache; switching the option off restores the bf16 prefill bit for bit."""
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    from vz_hip import binding as B
    O, sd, synth, cfg = env["O"], env["sd"], env["synth"], env["cfg"]
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=1, max_text=64, weight_fp8=True)
    eng = model.engine
    sdq = O.quantize_state_dict(sd)
    ids = synth.synth_ids(150, cfg.vocab, image_pos=-1, seed=19)
    emb = eng.embed_tokens(ids).unsqueeze(0)
    bf16_prefill, _ = eng.prefill(emb, [150], all_logits=True, last_logits=False)
    eng.set_prefill_fp8(True)
    short, _ = eng.prefill(emb, [150], all_logits=True, last_logits=False)
    assert torch.equal(short, bf16_prefill)            # below 768 prefill rows the option keeps the bf16 GEMMs (not worth the quantiser launches)
    B.check(B.lib().vz_tune_set(22, 0))               # ... unless told otherwise: this test runs the fp8 path on 150 rows
    full, _ = eng.prefill(emb, [150], all_logits=True, last_logits=False)
    lo_bf, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.BF16), P=O.BF16_FP8ACT)
    lo_32, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32_FP8ACT)
    band("fp8 MFMA prefill logits", full, lo_bf, lo_32)
    # what e4m3 activations cost on these (random, worst-case) weights: the distance to the weight-only-quantised model is of the size of
    # the band itself - two identically quantising implementations already differ by it (a bf16-level difference flips e4m3 roundings)
    lo_wonly, _ = O.llm_forward(cfg, sdq, O.embed_tokens(sd, ids.unsqueeze(0), O.FP32), P=O.FP32)
    e_act = errs(full, lo_wonly)[1]
    record("fp8 MFMA prefill: cost of e4m3 activations", hip_vs_weight_only_oracle=e_act, hip_vs_bf16_prefill_same_engine=errs(full, bf16_prefill)[1],
           oracle_fp8act_vs_weight_only=errs(lo_32, lo_wonly)[1])
    assert e_act <= 2.0 * errs(lo_32, lo_wonly)[1] + 5e-3
    # decode continues from the fp8-prefilled cache (bf16 activations from here on): close to the prefill row of the next position
    S0 = 120
    _, last = eng.prefill(emb[:, :S0].contiguous(), [S0], all_logits=False, last_logits=True)
    eng.decode_begin(ids[S0:S0 + 1].to(torch.int32), [S0], [S0])
    _, lg = eng.decode_steps(1, return_logits=True)
    e_or = errs(lo_bf, lo_32)[1]
    check_close("decode step after an fp8 prefill vs the fp8 prefill row", lg[0, 0], full[0, S0], 0.25, 3.0 * e_or + 2e-2)
    out = model.generate(input_ids=ids[:64].unsqueeze(0), do_sample=False, max_new_tokens=5, eos_token_id=None, pad_token_id=2)
    assert out.shape == (1, 5)
    eng.set_prefill_fp8(False)
    B.check(B.lib().vz_tune_set(22, 768))
    again, _ = eng.prefill(emb, [150], all_logits=True, last_logits=False)
    assert torch.equal(again, bf16_prefill)
    del model
    torch.cuda.empty_cache()
