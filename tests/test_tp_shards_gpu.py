"""Tensor-parallel shards on the real kernels (one GPU): two engines are built with tp_size=2 (rank 0 / rank 1); their
packed weight shards drive the same HIP operators the engine launches, with the all-reduce emulated by a bf16 sum, and
the result must equal the tp_size=1 engine's prefill logits.  Pins the shard packing (stacked local q|k|v, per-shard
gate/up interleave, column slices of o/down, zero-padded vocab shards) and the local-head attention shapes; the RCCL
calls themselves need >= 2 GPUs and are covered by the driver's multi-GPU run."""
import pytest
import torch

from util import check_close

pytestmark = pytest.mark.gpu


def test_tp2_shards_reproduce_single_gpu_prefill():
    from vz_hip import binding as B, synth
    from vz_hip.engine import Engine, rope_tables
    cfg = synth.ArchConfig(n_layers=1, vocab=1001, clip_layers=20)      # odd vocab: ragged vocab shards
    full = Engine(cfg, max_ctx=128, max_tiles=1, max_text=8)
    full.load_synthetic(0)
    shards = []
    for r in range(2):
        e = Engine(cfg, max_ctx=128, max_tiles=1, max_text=8, tp_size=2, tp_rank=r)
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
    Hq, Hkv = cfg.n_heads // 2, cfg.n_kv_heads // 2

    def allreduce(parts):
        return sum(p.float() for p in parts).bfloat16()

    xs = x
    parts_o, parts_d, atts = [], [], []
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
        assert act.shape[1] == cfg.inter // 2
        parts_d.append(B.linear(act, w["llm.0.down.w"], residual=xs if r == 0 else None))
    xs = allreduce(parts_d)
    logits = []
    for r, e in enumerate(shards):
        w = e.w
        h = B.rmsnorm(xs, w["llm.norm"], cfg.rms_eps)
        assert w["llm.lm_head"].shape[0] == 501
        logits.append(B.linear(h, w["llm.lm_head"], out_fp32=True))
    got = torch.cat(logits, dim=1)[:, :cfg.vocab]
    assert float(logits[1][:, 500:].abs().max()) == 0.0          # the padded vocab row of the last shard is inert
    check_close("tp2 shards vs tp1 engine logits", got, ref[0], 2e-2, 6e-3)
    for e in shards + [full]:
        e.close()
