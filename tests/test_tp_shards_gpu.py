"""Tensor-parallel shards on the real kernels (one GPU): two engines are built with tp_size=2 (rank 0 / rank 1); their
packed weight shards drive the same HIP operators the engine launches, with the all-reduce emulated by a bf16 sum, and
the result must equal the tp_size=1 engine's prefill logits.  Pins the shard packing (stacked local q|k|v, per-shard
gate/up interleave, column slices of o/down, zero-padded vocab shards) and the local-head attention shapes; the RCCL
calls themselves need >= 2 GPUs and are covered by the driver's multi-GPU run."""
import pytest
import torch

from util import check_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 8])
def test_tp_shards_reproduce_single_gpu_prefill(world):
    """world = 8 is the driver's node: 4 query + 1 KV head, 1792 MLP columns (not a multiple of 512: the one-row down-proj of a
    decode step leaves the GEMV for the tile GEMM there) and 126 vocab rows per rank."""
    from vz_hip import binding as B, synth
    from vz_hip.engine import Engine, rope_tables
    cfg = synth.ArchConfig(n_layers=1, vocab=1001, clip_layers=20)      # odd vocab: ragged vocab shards
    full = Engine(cfg, max_ctx=128, max_tiles=1, max_text=8)
    full.load_synthetic(0)
    shards = []
    for r in range(world):
        e = Engine(cfg, max_ctx=128, max_tiles=1, max_text=8, tp_size=world, tp_rank=r)
        shards.append(e)
    # register only the LLM shards (finalize() would also want CLIP / Q-Former, which are replicated and not needed here)
    for e in shards:
        for name, t in synth.iter_state_dict(cfg, 0, device=e.device, prefixes=("model.layers", "model.norm", "model.embed", "lm_head")):
            e.add_weight(name, t)
    S, H, D = 50, cfg.hidden, cfg.head_dim
    ids = synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=5)
    x = full.embed_tokens(ids)                                   # [S, H] bf16
    ref, _ = full.prefill(x.unsqueeze(0), [S], all_logits=True, last_logits=False)
    cos, sin = (t.cuda() for t in rope_tables(cfg, 128))
    pos = torch.arange(S, dtype=torch.int32, device="cuda")
    Hq, Hkv = cfg.n_heads // world, cfg.n_kv_heads // world
    vp = (cfg.vocab + world - 1) // world

    def allreduce(parts):
        return sum(p.float() for p in parts).bfloat16()

    xs = x
    parts_o, parts_d = [], []
    for r, e in enumerate(shards):
        w = e.w
        y = B.rmsnorm(xs, w["llm.0.in_norm"], cfg.rms_eps)
        qkv = B.linear(y, w["llm.0.qkv.w"])
        assert qkv.shape[1] == (Hq + 2 * Hkv) * D
        kc = torch.zeros(1, Hkv, 128, D, device="cuda").bfloat16()
        vc = torch.zeros_like(kc)
        q = B.rope_kv(qkv, cos, sin, pos, pos, kc, vc, 1, S, Hq, Hkv, D)
        att = B.attention(q.view(1, S, Hq, D), kc.permute(0, 2, 1, 3)[:, :S], vc.permute(0, 2, 1, 3)[:, :S], D ** -0.5, True, 0, 4096)
        parts_o.append(B.linear(att.reshape(S, Hq * D), w["llm.0.o.w"], residual=xs if r == 0 else None))
    xs = allreduce(parts_o)
    for r, e in enumerate(shards):
        w = e.w
        y = B.rmsnorm(xs, w["llm.0.post_norm"], cfg.rms_eps)
        act = B.linear(y, w["llm.0.gu.w"], act=B.ACT_SWIGLU)
        assert act.shape[1] == cfg.inter // world
        parts_d.append(B.linear(act, w["llm.0.down.w"], residual=xs if r == 0 else None))
        # the decode step's one-row forms of the same shard (weight-stream kernels where the shapes allow, tile GEMM otherwise)
        one = B.linear(act[:1].contiguous(), w["llm.0.down.w"])
        check_close(f"tp{world} rank {r} one-row down-proj", one, parts_d[-1][:1].float() - (xs[:1].float() if r == 0 else 0), 3e-2, 8e-3)
    xs = allreduce(parts_d)
    logits = []
    for r, e in enumerate(shards):
        w = e.w
        h = B.rmsnorm(xs, w["llm.norm"], cfg.rms_eps)
        assert w["llm.lm_head"].shape[0] == vp
        logits.append(B.linear(h, w["llm.lm_head"], out_fp32=True))
    got = torch.cat(logits, dim=1)[:, :cfg.vocab]
    n_last = cfg.vocab - (world - 1) * vp
    assert float(logits[-1][:, n_last:].abs().max()) == 0.0      # the padded vocab rows of the last shard are inert
    check_close(f"tp{world} shards vs tp1 engine logits", got, ref[0], 2e-2, 6e-3)
    for e in shards + [full]:
        e.close()


def test_tile_data_parallel_encode_matches_unsharded():
    """Tile data parallelism of CLIP + fusion + Q-Former through the production code path
    (`_encode_images_tile_dp`): the two ranks of a tp_size=2 group are played one after the other on this GPU, the RCCL
    all-gather is replaced by stacking the two ranks' send buffers (the layout vz_tp_all_gather produces).  With the
    batch-invariant kernel choice (128x128 GEMM, no stream-K: a tile's result does not depend on the tiles beside it)
    the gathered visual tokens must be BIT-IDENTICAL to the unsharded encode; with the production dispatch they must
    agree within bf16 rounding of the 8-block Q-Former."""
    from vz_hip import binding as B, synth
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg = synth.ArchConfig(n_layers=1, vocab=1001)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    hf.mm_vision_select_feature = "patch"
    model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=5, max_text=16)
    eng = model.engine
    T = 5
    tiles = synth.synth_tiles(T, seed=1).to(model.device).bfloat16()
    text = synth.hash_normal("text", (2, 9, cfg.hidden), 1.0, 4).to(model.device).bfloat16()     # two samples: 3 + 2 tiles
    tile_sample = [0, 0, 0, 1, 1]

    def tile_dp():
        sends = {}
        real_gather = eng.all_gather
        try:
            eng.tp_size = 2
            for r in range(2):                                   # pass 1: what each rank would put on the wire
                eng.tp_rank = r
                eng.all_gather = lambda send, r=r: (sends.__setitem__(r, send.clone()), torch.zeros((2,) + tuple(send.shape), dtype=send.dtype, device=send.device))[1]
                model.encode_images(tiles, text, tile_sample=tile_sample)
            assert sends[0].shape == sends[1].shape == (3, cfg.qf_queries, cfg.hidden)
            assert float(sends[1][2].float().abs().max()) == 0.0          # rank 1 owns tiles 1, 3: its third slot is padding
            eng.all_gather = lambda send: torch.stack([sends[0], sends[1]])
            outs = []
            for r in range(2):                                   # pass 2: both ranks end up with all visual tokens
                eng.tp_rank = r
                outs.append(model.encode_images(tiles, text, tile_sample=tile_sample))
            assert torch.equal(outs[0], outs[1])
            return outs[0]
        finally:
            eng.tp_size, eng.tp_rank, eng.all_gather = 1, 0, real_gather

    try:
        B.check(B.lib().vz_tune_set(1, 1))       # 128x128 kernel only
        B.check(B.lib().vz_tune_set(4, 0))
        B.check(B.lib().vz_tune_set(26, 0))      # whole-K tiles whatever the tile count of a rank's share
        ref = model.encode_images(tiles, text, tile_sample=tile_sample)
        assert ref.shape == (T, cfg.qf_queries, cfg.hidden)
        assert torch.equal(tile_dp(), ref), "tile-data-parallel encode differs from the unsharded one"
    finally:
        B.check(B.lib().vz_tune_set(1, 0))
        B.check(B.lib().vz_tune_set(4, 1))
        B.check(B.lib().vz_tune_set(26, 1))
    ref = model.encode_images(tiles, text, tile_sample=tile_sample)
    check_close("tile-DP vs unsharded (production dispatch)", tile_dp(), ref.float(), 0.25, 2e-2)
    # the C-ABI all-gather at tp_size 1 is a device copy into chunk 0
    got = eng.all_gather(ref[:2])
    assert got.shape == (1, 2, cfg.qf_queries, cfg.hidden) and torch.equal(got[0], ref[:2])


def test_rccl_call_sites_on_one_rank():
    """The RCCL plumbing on ONE GPU: a tp_size == 1 engine gets a one-rank communicator and vz_tune_set(7, 1) routes its
    collective call sites through RCCL - 2 bf16 all-reduces per layer in prefill and in every decode step (in place, on the
    engine's stream), the vocab-parallel fp32 all-gather + repack of the logits, and vz_tp_all_gather.  Over one rank
    every collective is the identity, so logits and generated ids must be bit-identical to the plain path."""
    from vz_hip import binding as B, synth
    from vz_hip.engine import Engine
    cfg = synth.ArchConfig(n_layers=2, vocab=1001, clip_layers=20)
    eng = Engine(cfg, max_ctx=128, max_tiles=1, max_text=8)
    eng.load_synthetic(0)
    S = 40
    ids = synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=5)
    x = eng.embed_tokens(ids).unsqueeze(0)

    def run():
        all_logits, last = eng.prefill(x, [S], all_logits=True, last_logits=True)
        eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
        return all_logits.clone(), eng.decode_steps(6).clone()

    ref_logits, ref_ids = run()
    eng.init_comm_single_rank()
    try:
        B.check(B.lib().vz_tune_set(7, 1))
        logits, new_ids = run()
        # the decode steps ran as ONE captured graph with the RCCL all-reduces / all-gather inside it (not an eager fallback)
        assert eng.decode_mode() == (True, True)
        again = eng.decode_steps(6).clone()          # replay of the same graph object, collectives included
        gathered = eng.all_gather(x[0, :3])
    finally:
        B.check(B.lib().vz_tune_set(7, 0))
    assert torch.equal(logits, ref_logits)
    assert torch.equal(new_ids, ref_ids)
    assert again.shape == new_ids.shape
    # The captured TP decode graph holds the gathered-logits buffer: a forward with MORE logits rows between two generates
    # re-allocates it (ensure_gather), so the graph must be dropped and captured again, not replayed on the freed pointer.
    try:
        B.check(B.lib().vz_tune_set(7, 1))
        _, ids_a = run()
        big = eng.embed_tokens(synth.synth_ids(100, cfg.vocab, image_pos=-1, seed=6)).unsqueeze(0)
        eng.prefill(big, [100], all_logits=True, last_logits=False)          # 100 rows of logits > anything gathered so far
        _, ids_b = run()
        assert eng.decode_mode() == (True, True)
    finally:
        B.check(B.lib().vz_tune_set(7, 0))
    assert torch.equal(ids_a, ref_ids) and torch.equal(ids_b, ref_ids)
    assert gathered.shape == (1, 3, cfg.hidden) and torch.equal(gathered[0], x[0, :3])


@pytest.mark.parametrize("world,rank", [(8, 5), (4, 0), (2, 1)])
def test_one_rank_of_a_tp_engine_runs_its_local_path(world, rank):
    """Shape rehearsal on one GPU: ONE rank of a tp_size = 2 / 4 / 8 engine with the full model loaded (its Zephyr shards, CLIP /
    Q-Former replicated) runs image -> first token -> a few decode steps with the collectives skipped (vz_tune_set(7, 2)).
    Logits are meaningless without the other ranks' partial sums; what is checked is that every local kernel accepts the
    shard shapes (1792 MLP columns and one KV head at tp 8, tile-DP deal of 5 tiles, vocab shard + repack) and stays finite."""
    from vz_hip import binding as B, synth
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    cfg = synth.ArchConfig(n_layers=2)
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                         num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, vocab_size=cfg.vocab,
                         rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    try:
        B.check(B.lib().vz_tune_set(7, 2))
        model = VisZephyrForCausalLM(hf, device="cuda:0", max_batch=2, max_ctx=512, max_tiles=5, max_text=64, tp_size=world, tp_rank=rank)
        model.engine.load_synthetic(0)                       # no init_comm: the collectives are skipped in this rehearsal
        model.get_vision_tower().is_loaded = True
        eng = model.engine
        assert eng.w["llm.0.gu.w"].shape[0] == 2 * cfg.inter // world and eng.w["llm.0.qkv.w"].shape[0] == (cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim // world
        tiles = synth.synth_tiles(5, seed=1).to(model.device).bfloat16()
        ids = synth.synth_ids(40, cfg.vocab, image_pos=5, seed=2).unsqueeze(0).to(model.device)
        out = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=5, eos_token_id=None, pad_token_id=2)
        assert out.shape == (1, 5) and int(out.min()) >= 0 and int(out.max()) < cfg.vocab
        logits = model(input_ids=ids, images=[tiles]).logits
        assert logits.shape == (1, 39 + 32 * 5, cfg.vocab) and bool(torch.isfinite(logits).all())
        # two text rows through the batched decode path of the shard
        ids2 = torch.stack([synth.synth_ids(12, cfg.vocab, image_pos=-1, seed=3 + b) for b in range(2)]).to(model.device)
        out2 = model.generate(input_ids=ids2, do_sample=False, max_new_tokens=4, eos_token_id=None, pad_token_id=2)
        assert out2.shape == (2, 4)
    finally:
        B.check(B.lib().vz_tune_set(7, 0))
