"""World-size-2 check of the tensor-parallel sharding plan (vz_hip/tp.py) on CPU with gloo.

Each rank takes its slice of every Zephyr weight, runs the oracle's own arithmetic on its heads / MLP
columns, all-reduces the row-parallel partial sums and all-gathers the vocab-parallel logits; the result
must equal the unsharded oracle.  This pins the plan the RCCL engine will execute (the collectives'
positions and the slice boundaries) without a GPU."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_q):
    for p in (REPO, os.path.join(REPO, "vision-zephyr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vz_oracle as O
        from vz_hip import synth, tp
        torch.set_num_threads(2)
        cfg = synth.ArchConfig(hidden=512, inter=1024, n_layers=2, n_heads=8, n_kv_heads=4, head_dim=64, vocab=1001)
        tp.check_divisible(cfg, world)
        sd = synth.state_dict(cfg, 0, prefixes=("model.layers", "model.norm", "model.embed", "lm_head"))
        mine = {k: tp.shard(k, v, rank, world) for k, v in sd.items()}
        ids = synth.synth_ids(12, cfg.vocab, image_pos=-1, seed=3).unsqueeze(0)
        x = O.embed_tokens(sd, ids, O.FP32)
        B, S, H = x.shape
        nh, nkv, hd = cfg.n_heads // world, cfg.n_kv_heads // world, cfg.head_dim
        pos = torch.arange(S).unsqueeze(0)
        cos, sin = O.rope_tables(cfg, pos)
        cos, sin = cos.unsqueeze(2), sin.unsqueeze(2)
        keep = (torch.arange(S).view(1, 1, 1, S) <= torch.arange(S).view(1, 1, S, 1))
        n_allreduce = 0
        for i in range(cfg.n_layers):
            p = f"model.layers.{i}."
            y = O.rmsnorm(x, mine[p + "input_layernorm.weight"], cfg.rms_eps)
            q = (y @ mine[p + "self_attn.q_proj.weight"].t()).view(B, S, nh, hd)
            k = (y @ mine[p + "self_attn.k_proj.weight"].t()).view(B, S, nkv, hd)
            v = (y @ mine[p + "self_attn.v_proj.weight"].t()).view(B, S, nkv, hd)
            q = q * cos + O._rot_half(q) * sin
            k = k * cos + O._rot_half(k) * sin
            a = O._attention(q, k, v, hd ** -0.5, O.FP32, mask=keep).reshape(B, S, nh * hd)
            part = a @ mine[p + "self_attn.o_proj.weight"].t()          # row-parallel partial sum
            dist.all_reduce(part)
            n_allreduce += 1
            x = x + part
            y = O.rmsnorm(x, mine[p + "post_attention_layernorm.weight"], cfg.rms_eps)
            g = y @ mine[p + "mlp.gate_proj.weight"].t()
            u = y @ mine[p + "mlp.up_proj.weight"].t()
            part = (torch.nn.functional.silu(g) * u) @ mine[p + "mlp.down_proj.weight"].t()
            dist.all_reduce(part)
            n_allreduce += 1
            x = x + part
        h = O.rmsnorm(x, mine["model.norm.weight"], cfg.rms_eps)
        local = h @ mine["lm_head.weight"].t()
        lo, hi = tp.vocab_range(cfg.vocab, rank, world)
        assert local.shape[-1] == hi - lo
        pad = (cfg.vocab + world - 1) // world
        buf = torch.zeros(B, S, pad)
        buf[..., :hi - lo] = local
        gathered = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        logits = torch.cat([gathered[r][..., :tp.vocab_range(cfg.vocab, r, world)[1] - tp.vocab_range(cfg.vocab, r, world)[0]]
                            for r in range(world)], dim=-1)
        ref, _ = O.llm_forward(cfg, sd, O.embed_tokens(sd, ids, O.FP32))
        err = float((logits - ref).abs().max() / ref.abs().max())
        out_q.put((rank, err, n_allreduce, tp.allreduce_bytes_per_layer(cfg, S)))
    finally:
        dist.destroy_process_group()


def _tile_dp_worker(rank, world, port, out_q):
    """tile data parallelism of CLIP + fusion + Q-Former: the plan of vz_hip/tp.py with the oracle as the encoder."""
    for p in (REPO, os.path.join(REPO, "vision-zephyr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vz_oracle as O
        from vz_hip import synth, tp
        torch.set_num_threads(2)
        # a narrow model of the same shape: 24 CLIP layers (the fusion reads the last 21 hidden states), 8 Q-Former blocks
        cfg = synth.ArchConfig(hidden=128, clip_hidden=64, clip_inter=128, clip_heads=4, clip_image=56, qf_heads=4, qf_kv_dim=320)
        sd = synth.state_dict(cfg, 0, prefixes=("model.vision_tower", "model.mm_projector"))
        worst = 0.0
        for T in (1, 3, 5):                       # fewer tiles than ranks, odd counts, the 5-tile anyres case
            tiles = synth.synth_tiles(T, seed=7, size=cfg.clip_image)
            text = synth.hash_normal("text", (T, 6, cfg.hidden), 1.0, 4)
            mine = tp.local_tiles(T, rank, world)
            per = tp.tiles_per_rank(T, world)
            send = torch.zeros(per, cfg.qf_queries, cfg.hidden)
            if mine:
                send[:len(mine)] = O.encode_images(cfg, sd, tiles[mine], text[mine], O.FP32)
            got = [torch.zeros_like(send) for _ in range(world)]
            dist.all_gather(got, send)
            out = torch.cat(got, 0)[tp.gathered_index(T, world)]
            ref = O.encode_images(cfg, sd, tiles, text, O.FP32)
            assert out.shape == ref.shape
            worst = max(worst, float((out - ref).abs().max() / ref.abs().max()))
        out_q.put((rank, worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tile_data_parallel_plan_matches_unsharded_oracle():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_tile_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-5, (rank, err)


@pytest.mark.timeout(300)
def test_tp2_sharding_plan_matches_unsharded_oracle():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, n_ar, nbytes in res:
        assert err < 1e-5, (rank, err)
        assert n_ar == 4 and nbytes == 2 * 12 * 512 * 2


def test_shard_kinds_and_ranges():
    for p in (REPO, os.path.join(REPO, "vision-zephyr_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from vz_hip import synth, tp
    assert tp.kind_of("model.layers.3.self_attn.q_proj.weight") == tp.COL
    assert tp.kind_of("model.layers.3.mlp.down_proj.weight") == tp.ROW
    assert tp.kind_of("lm_head.weight") == tp.VOCAB
    assert tp.kind_of("model.mm_projector.blocks.0.ffn.0.weight") == tp.REPL
    assert tp.kind_of("model.layers.0.input_layernorm.weight") == tp.REPL
    cfg = synth.ArchConfig()
    for w in (1, 2, 4, 8):
        tp.check_divisible(cfg, w)
        spans = [tp.vocab_range(32001, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == 32001 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    with pytest.raises(ValueError):
        tp.check_divisible(cfg, 3)
    for T in (1, 4, 5, 64):
        for w in (1, 2, 4, 8):
            owned = sorted(t for r in range(w) for t in tp.local_tiles(T, r, w))
            assert owned == list(range(T))
            per = tp.tiles_per_rank(T, w)
            assert all(len(tp.local_tiles(T, r, w)) <= per for r in range(w))
            gi = tp.gathered_index(T, w).tolist()
            assert gi == [(t % w) * per + t // w for t in range(T)] and len(set(gi)) == T
    t = torch.arange(8 * 6).view(8, 6)
    assert torch.equal(torch.cat([tp.shard("model.layers.0.mlp.up_proj.weight", t, r, 2) for r in range(2)], 0), t)
    assert torch.equal(torch.cat([tp.shard("model.layers.0.self_attn.o_proj.weight", t, r, 2) for r in range(2)], 1), t)
