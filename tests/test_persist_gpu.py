"""The persistent decode-token kernel (csrc/decode_persist.hip: batch-1 decode steps as ONE resident grid per token, the 161 launches
of a step as phases with in-launch hand-offs) against the launch chain (the default; the persistent kernel is opt-in, vz_tune_set(28, 1): it measured slower on MI355X): the same arithmetic in the
same order, so logits and ids must be EQUAL bit for bit - eager and through the captured per-token graph, at 2 and at 32 layers, short
and long contexts (1 .. 18 context splits of the attention phase) - and its bounded waits must end the launch and raise
VZ_ASYNC_PERSIST instead of hanging (hf:models/mistral/modeling_mistral.py:202-241 is the arithmetic of a step; the oracle / reference
parity of the launch chain is tests/test_stages_gpu.py / test_depth32_gpu.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(n_layers, max_ctx, weight_fp8=False, max_batch=1):
    from vz_hip import synth
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg = synth.ArchConfig(n_layers=n_layers)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=n_layers, num_attention_heads=cfg.n_heads,
                         num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                         sliding_window=cfg.sliding_window, eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    return cfg, synth, VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=max_batch, max_ctx=max_ctx, max_tiles=1, max_text=max_ctx, weight_fp8=weight_fp8)


def _knob(v):
    from vz_hip import binding as B
    B.check(B.lib().vz_tune_set(28, int(v)))


def _steps(model, emb, n, persist):
    """prefill + n eager decode steps with their logits; (ids [n], logits [n, V], ran on the persistent kernel?)"""
    eng = model.engine
    _knob(persist)
    try:
        S = emb.shape[1]
        _, last = eng.prefill(emb, [S], all_logits=False, last_logits=True)
        eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
        ids, lg = eng.decode_steps(n, return_logits=True)
        torch.cuda.synchronize()
        eng.check_async()
        return ids[0].clone(), lg[:, 0].clone(), eng.persist_mode()
    finally:
        _knob(0)


@pytest.fixture(scope="module")
def small():
    cfg, synth, model = _model(2, 2300)
    if not model.engine.lib.vz_abi_version():
        pytest.skip("no library")
    yield cfg, synth, model
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("S", [5, 127, 300, 1100, 2200])
def test_persistent_steps_equal_the_launch_chain_bit_for_bit(small, S):
    cfg, synth, model = small
    emb = model.engine.embed_tokens(synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=900 + S).unsqueeze(0))
    ids_p, lg_p, mode_p = _steps(model, emb, 6, persist=1)
    ids_l, lg_l, mode_l = _steps(model, emb, 6, persist=0)
    if not mode_p:
        pytest.skip("the persistent kernel is not available on this device (needs 256 CUs)")
    assert mode_p and not mode_l
    assert torch.isfinite(lg_p).all()
    assert torch.equal(ids_p, ids_l), (ids_p.tolist(), ids_l.tolist())
    assert torch.equal(lg_p, lg_l), f"S={S}: max |diff| {float((lg_p - lg_l).abs().max()):.3e}"


def test_persistent_graph_replay_equals_launch_chain_and_sampling_tail(small):
    """generate() through the captured per-token graph (one kernel node + the tail per token) = the launch chain's ids; with the
    device-side sampler as the tail the draws are equal too (same logits, same Philox counters)."""
    cfg, synth, model = small
    ids = synth.synth_ids(200, cfg.vocab, image_pos=-1, seed=77).unsqueeze(0)
    outs = {}
    for persist in (1, 0):
        _knob(persist)
        try:
            g = model.generate(input_ids=ids, do_sample=False, max_new_tokens=40, eos_token_id=None)[0].tolist()
            mode = model.engine.persist_mode()
            s_ = model.generate(input_ids=ids, do_sample=True, temperature=0.7, top_k=50, max_new_tokens=24, eos_token_id=None, seed=5)[0].tolist()
            model.engine.check_async()
            outs[persist] = (g, s_, mode)
        finally:
            _knob(0)
    if not outs[1][2]:
        pytest.skip("the persistent kernel is not available on this device")
    assert outs[1][0] == outs[0][0] and outs[1][1] == outs[0][1]
    assert model.engine.decode_mode()[0] or os.environ.get("VZ_NO_GRAPH")


def test_expired_hand_off_ends_the_launch_and_raises(small):
    """a lagging arrival counter (a workgroup that never arrives, as far as the others can tell): every wait is bounded, the launch
    ends, the async error word reads VZ_ASYNC_PERSIST, and the next call - which zeroes the counters - is clean again.  Same for a
    preset abort word (the fast path every workgroup takes once one of them has given up)."""
    from vz_hip import binding as B
    cfg, synth, model = small
    eng = model.engine
    emb = eng.embed_tokens(synth.synth_ids(64, cfg.vocab, image_pos=-1, seed=5).unsqueeze(0))
    ids_ref, lg_ref, mode = _steps(model, emb, 2, persist=1)
    if not mode:
        pytest.skip("the persistent kernel is not available on this device")
    for word, value in ((0, -1000), (8, 1)):
        S = emb.shape[1]
        _knob(1)
        try:
            _, last = eng.prefill(emb, [S], all_logits=False, last_logits=True)
            eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
            eng.persist_poke(word, value)
            eng.decode_steps(1)
            torch.cuda.synchronize()                       # the launch ended (bounded waits): nothing hangs
        finally:
            _knob(0)
        with pytest.raises(RuntimeError, match="persistent decode-token kernel"):
            eng.check_async()
        ids2, lg2, _ = _steps(model, emb, 2, persist=1)
        assert torch.equal(ids2, ids_ref) and torch.equal(lg2, lg_ref)
    assert B.VZ_ASYNC_PERSIST == 3


@pytest.fixture(scope="module")
def small_fp8():
    cfg, synth, model = _model(2, 2300, weight_fp8=True)
    yield cfg, synth, model
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("S", [5, 300, 2200])
def test_fused_attention_o_launch_on_e4m3_weights(small_fp8, S):
    """the e4m3 instantiation (W8A16 engines: the O projection's 1-byte rows + per-row scales, gemv_bf16_kernel<.., FP8>'s chunk order) = the two launches"""
    test_fused_attention_o_launch_equals_the_two_launches(small_fp8, S)


@pytest.mark.parametrize("S", [5, 127, 300, 1100, 2200])
def test_fused_attention_o_launch_equals_the_two_launches(small, S):
    """csrc/attn_o_fused.hip (the default batch-1 route: the O projection's workgroups ride in the decode attention's grid and stream
    their weights under its latency chain) against vz_tune_set(30, 0) (attention kernel, then the O projection's GEMV): the same
    arithmetic in the same order - logits and ids EQUAL bit for bit, eager and through the captured graph."""
    from vz_hip import binding as B
    cfg, synth, model = small
    eng = model.engine
    emb = eng.embed_tokens(synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=700 + S).unsqueeze(0))
    res = {}
    for fused in (1, 0):
        B.check(B.lib().vz_tune_set(30, fused))
        try:
            _, last = eng.prefill(emb, [S], all_logits=False, last_logits=True)
            eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
            ids, lg = eng.decode_steps(5, return_logits=True)
            _, last = eng.prefill(emb, [S], all_logits=False, last_logits=True)
            eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
            ids_g = eng.decode_steps(12)
            torch.cuda.synchronize()
            eng.check_async()
            res[fused] = (ids[0].clone(), lg[:, 0].clone(), ids_g[0].clone())
        finally:
            B.check(B.lib().vz_tune_set(30, 1))
    for fused in (1,):
        assert torch.isfinite(res[fused][1]).all()
        assert torch.equal(res[fused][0], res[0][0]) and torch.equal(res[fused][2], res[0][2]), fused
        assert torch.equal(res[fused][1], res[0][1]), f"knob {fused}, S={S}: max |diff| {float((res[fused][1] - res[0][1]).abs().max()):.3e}"


@pytest.fixture(scope="module")
def small2():
    cfg, synth, model = _model(2, 2300, max_batch=2)
    yield cfg, synth, model
    del model
    torch.cuda.empty_cache()


@pytest.mark.parametrize("S,short", [(40, 0), (700, 9), (1900, 300)])
def test_fused_attention_o_launch_with_two_rows(small2, S, short):
    """the same launch for a 2-row decode step (the rows' attention workgroups + the O projection's 512 workgroups with both rows' outputs in LDS;
    rows x context splits <= 32 keeps every workgroup resident): logits and ids of both rows EQUAL the two-launch route's, ragged lengths included"""
    from vz_hip import binding as B
    cfg, synth, model = small2
    eng = model.engine
    lens = [S, S - short]
    ids = torch.stack([synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=800 + S + r) for r in range(2)])
    emb = eng.embed_tokens(ids.reshape(-1)).view(2, S, -1)
    res = {}
    for fused in (1, 0):
        B.check(B.lib().vz_tune_set(30, fused))
        try:
            _, last = eng.prefill(emb, lens, all_logits=False, last_logits=True)
            eng.decode_begin(last.argmax(-1).to(torch.int32), lens, lens)
            toks, lg = eng.decode_steps(4, return_logits=True)
            _, last = eng.prefill(emb, lens, all_logits=False, last_logits=True)
            eng.decode_begin(last.argmax(-1).to(torch.int32), lens, lens)
            toks_g = eng.decode_steps(9)
            torch.cuda.synchronize()
            eng.check_async()
            res[fused] = (toks.clone(), lg.clone(), toks_g.clone())
        finally:
            B.check(B.lib().vz_tune_set(30, 1))
    assert torch.isfinite(res[1][1]).all()
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][2], res[0][2])
    assert torch.equal(res[1][1], res[0][1]), f"S={S}: max |diff| {float((res[1][1] - res[0][1]).abs().max()):.3e}"
