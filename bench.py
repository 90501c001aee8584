#!/usr/bin/env python
"""bench.py - image->first-token latency + decode tokens/s of the Vision-Zephyr hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): Zephyr-7B-beta
(32 layers, random-init hash weights, bf16) + CLIP ViT-L/14-336 + Q-Former, ONE request of 5 anyres
tiles (4 crops + 1 global) and a 1889-id prompt -> spliced sequence S = 2048, 128 greedy new tokens.
One "step" = one full `VisZephyrForCausalLM.generate` on that request, inputs resident in HBM:
  image->first-token  = tiles+ids -> CLIP -> fusion -> Q-Former -> splice -> 32-layer prefill -> argmax
  decode              = 127 further tokens against the KV cache.
`value` = decode tokens/s (whole job); `image_to_first_token_ms` rides in the same JSON line.
N > 1: one process per GPU, each serving its own request (the reference's own multi-GPU inference
scheme, ref:script/eval/eval_qa.sh:21-47) - replicas, no data-path collective, weak scaling.

Extra objects: `roofline` (the dominant kernel by time: the decode weight-streaming GEMV, HBM-bound;
HIP events on the launch stream in an instrumented replay of decode steps right after the timed
region), `roofline_prefill` (MFMA tile GEMM of the prefill, same method) and `cpu_baseline` (the CPU
oracle = port of the reference's fp32 CPU path, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


def build_model(n_layers, device, max_ctx, tp_size=1, tp_rank=0, weight_fp8=False):
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    hf = VisZephyrConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=n_layers, num_attention_heads=32,
                         num_key_value_heads=8, vocab_size=32000, rms_norm_eps=1e-5, sliding_window=4096,
                         eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    hf.image_aspect_ratio = "anyres"
    hf.mm_grid_pinpoints = "[[336, 672], [672, 336], [336, 1008], [1008, 336], [672, 672]]"
    hf.mm_hidden_size = 5120
    hf.mm_vision_select_layer = "-2,-5,-8,-11,6"
    return VisZephyrForCausalLM.from_synthetic(hf, seed=0, device=device, max_batch=1, max_ctx=max_ctx, max_tiles=5,
                                               max_text=2048, tp_size=tp_size, tp_rank=tp_rank, weight_fp8=weight_fp8)


def algorithmic_work(cfg, S, n_tiles):
    """SURVEY.md section 8(d) figures: bytes per decoded token (weights, bf16) and prefill linear FLOPs."""
    H, I, V, L = cfg.hidden, cfg.inter, cfg.vocab, cfg.n_layers
    qkv = (cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim
    per_layer_w = qkv * H + H * H + 2 * I * H + I * H
    decode_weight_bytes = 2 * (L * per_layer_w + V * H)                       # 14.22 GB at L=32 (lm_head incl., embedding row excl.)
    prefill_linear_flops = 2 * S * L * per_layer_w + 2 * 1 * V * H            # generate(): lm_head on the last row only
    return decode_weight_bytes, prefill_linear_flops


def pmc_traffic():
    """HBM read bytes per GEMV launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE in its own run, x1024 x2
    per the gfx950 correction); PMC counters cannot be read from inside this process."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_gemv_fetch.json")) as f:
            return int(json.load(f)["hbm_read_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(cfg_full, S, n_tiles, n_new):
    """The oracle (CPU port of the reference's fp32 path) on this box's host cores, bounded sample:
    1 CLIP tile, 1 Q-Former tile at the full text length, 1 decoder layer of prefill at S, 8 decode tokens
    over 2 layers at context S; per-layer / per-tile times are scaled to 32 layers / 5 tiles."""
    from oracle import vz_oracle as O
    from vz_hip import synth
    torch.set_grad_enabled(False)
    threads = torch.get_num_threads()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    cfg = cfg_full.small(n_layers=2)
    t0 = time.perf_counter()
    sd = {k: v.cpu() for k, v in synth.iter_state_dict(cfg, 0, device=dev)}
    t_w = time.perf_counter() - t0
    tiles = synth.synth_tiles(1, seed=1)
    L = S - 32 * n_tiles
    ids = synth.synth_ids(L, cfg.vocab, image_pos=-1, seed=2)
    te = O.embed_tokens(sd, ids, O.FP32).unsqueeze(0)
    t0 = time.perf_counter(); feats = O.clip_tower(cfg, sd, tiles); t_clip = time.perf_counter() - t0
    t0 = time.perf_counter(); O.qformer(cfg, sd, feats, te); t_qf = time.perf_counter() - t0
    emb = O.embed_tokens(sd, synth.synth_ids(S, cfg.vocab, image_pos=-1, seed=3), O.FP32).unsqueeze(0)
    cfg1 = cfg.small(n_layers=1)
    t0 = time.perf_counter(); O.llm_forward(cfg1, sd, emb, last_only=True); t_pre1 = time.perf_counter() - t0
    _, cache = O.llm_forward(cfg, sd, emb, last_only=True)
    x = emb[:, :1]
    n_dec = 8
    t0 = time.perf_counter()
    for _ in range(n_dec):
        m = torch.ones(1, cache.k[0].shape[1] + 1, dtype=torch.bool)
        _, cache = O.llm_forward(cfg, sd, x, attention_mask=m, cache=cache, last_only=True)
    t_dec2 = (time.perf_counter() - t0) / n_dec
    # lm_head alone (identical in the 1- and 2-layer runs): time it to split layer cost from head cost
    h = torch.randn(1, 1, cfg.hidden)
    t0 = time.perf_counter()
    for _ in range(4):
        O._lin(h, sd["lm_head.weight"], None, O.FP32)
    t_head = (time.perf_counter() - t0) / 4
    t_layer_dec = max(1e-9, (t_dec2 - t_head) / 2)
    tok_s = 1.0 / (cfg_full.n_layers * t_layer_dec + t_head)
    ttft = n_tiles * (t_clip + t_qf) + cfg_full.n_layers * (t_pre1 - t_head) + t_head
    return {"value": round(tok_s, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
            "image_to_first_token_ms": round(ttft * 1e3, 1),
            "sample": f"oracle fp32 on {threads} host threads ({os.cpu_count()} logical cpus): 1 CLIP tile {t_clip:.2f}s, "
                      f"1 Q-Former tile at L={L} {t_qf:.2f}s, 1 decoder layer prefill at S={S} {t_pre1:.2f}s, "
                      f"{n_dec} decode tokens over 2 layers at ctx {S} {t_dec2 * 1e3:.0f} ms/token, lm_head {t_head * 1e3:.0f} ms; "
                      f"scaled to {n_tiles} tiles / {cfg_full.n_layers} layers (weights generated in {t_w:.0f}s, untimed)"}


def fp8_leg_only(args):
    """child of the default run: the W8A16 engine on the same request; prints its own JSON object."""
    import torch
    from vz_hip import binding as B, synth
    device = "cuda:0"
    torch.cuda.set_device(device)
    n_tiles, n_ids, n_new = 5, 1889, args.new_tokens
    S = (n_ids - 1) + 32 * n_tiles
    model8 = build_model(args.layers, device, max_ctx=S + n_new + 16, weight_fp8=True)
    cfg = model8.arch
    tiles = synth.synth_tiles(n_tiles, seed=1).to(device, torch.bfloat16)
    ids = synth.synth_ids(n_ids, cfg.vocab, image_pos=5, seed=2).unsqueeze(0).to(device)

    def step8():
        tm = {}
        t0 = time.perf_counter()
        out = model8.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, eos_token_id=None,
                              pad_token_id=2, use_cache=True, timing=tm)
        torch.cuda.synchronize()
        assert out.shape == (1, n_new)
        return tm["t_first_token"] - t0, time.perf_counter() - tm["t_first_token"]
    step8()
    r8 = [step8() for _ in range(args.steps)]
    e8 = model8.engine
    emb8 = model8.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles])[4]
    _, last8 = e8.prefill(emb8, [S])
    e8.decode_begin(last8.argmax(-1).to(torch.int32), [S], [S])
    e8.prof_enable(True, B.K_GEMV)
    e8.decode_steps(16)
    torch.cuda.synchronize()
    n_l8, ms8 = e8.prof_read()
    e8.prof_enable(False)
    w_bytes8 = algorithmic_work(cfg, S, n_tiles)[0] // 2 + 4 * (cfg.n_layers * ((cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim
                                                                                 + 2 * cfg.hidden + 2 * cfg.inter) + cfg.vocab)
    ach8 = w_bytes8 / (4 * cfg.n_layers + 1) / (ms8 / max(1, n_l8) * 1e-3) / 1e9
    fp8_leg = {"value": round((n_new - 1) * args.steps / sum(b for _, b in r8), 2), "unit": "tokens/s",
               "image_to_first_token_ms": round(sum(a for a, _ in r8) / args.steps * 1e3, 2),
               "dtype": "w8a16: OCP e4m3 weights + per-row 2^e scales (decode stream), bf16 activations, bf16 MFMA prefill on the "
                        "dequantised weights",
               "roofline": {"bound": "hbm", "kernel": "gemv_bf16_kernel<.., FP8> (decode weight stream, 1 B per weight)",
                            "achieved": round(ach8, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach8 / HBM_PEAK_GBS, 4),
                            "traffic": None, "launches": n_l8, "avg_launch_ms": round(ms8 / max(1, n_l8), 5),
                            "algorithmic_bytes_per_launch": int(w_bytes8 / (4 * cfg.n_layers + 1))}}
    del model8, e8
    torch.cuda.empty_cache()
    print(json.dumps(fp8_leg), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=32, help="decoder layers (32 = Zephyr-7B; anything else is a debug run)")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp8-leg", action="store_true", help="skip the extra W8A16 (e4m3 weight stream) measurement")
    ap.add_argument("--fp8-leg-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.fp8_leg_only:
        return fp8_leg_only(args)

    rank = int(os.environ.get("RANK", "0"))
    tune = os.environ.get("VZ_TUNE", "")   # experiments only: "knob=value,..." for vz_tune_set; reported in config when set
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # rehearsal on a one-GPU box: VZ_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and uses gloo for the control-plane
    # collectives (barrier, max of the timings) - the multi-rank control flow runs, the GPUs are not what is measured
    single_dev = os.environ.get("VZ_BENCH_SINGLE_DEVICE", "0") == "1"
    if single_dev:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if single_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)

    from vz_hip import binding as B, synth
    for kv in filter(None, tune.split(",")):
        k, v = kv.split("=")
        B.check(B.lib().vz_tune_set(int(k), int(v)))
    n_tiles, n_ids, n_new = 5, 1889, args.new_tokens
    S = (n_ids - 1) + 32 * n_tiles
    # N > 1: replicas by default (one request per GPU, weak scaling).  VZ_BENCH_PARALLELISM=tp runs ONE request over a
    # tensor-parallel engine instead (Zephyr sharded over the N GPUs, RCCL all-reduce / all-gather; strong scaling).
    tp_mode = world > 1 and os.environ.get("VZ_BENCH_PARALLELISM", "replicas") == "tp"
    model = build_model(args.layers, device, max_ctx=S + n_new + 16, tp_size=world if tp_mode else 1,
                        tp_rank=rank if tp_mode else 0)
    eng, cfg = model.engine, model.arch
    tiles = synth.synth_tiles(n_tiles, seed=1).to(device, torch.bfloat16)
    ids = synth.synth_ids(n_ids, cfg.vocab, image_pos=5, seed=2).unsqueeze(0).to(device)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        tm = {}
        t0 = time.perf_counter()
        out = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, eos_token_id=None,
                             pad_token_id=2, use_cache=True, timing=tm)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        assert out.shape == (1, n_new)
        return tm["t_first_token"] - t0, t2 - tm["t_first_token"]

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    ttfts, decs = [], []
    for _ in range(args.steps):
        a, b = step()
        ttfts.append(a)
        decs.append(b)
    barrier()
    elapsed = time.perf_counter() - t_start
    stats = torch.tensor([elapsed, sum(ttfts), sum(decs)], dtype=torch.float64, device="cpu" if single_dev else device)
    if dist is not None:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    elapsed, ttft_sum, dec_sum = stats.tolist()

    # ---- roofline legs: instrumented replays of the same work, HIP events on the launch stream ----
    roof, roof_prefill = None, None
    if rank == 0 and not tp_mode:          # (a tensor-parallel engine needs every rank inside each collective)
        w_bytes, pre_flops = algorithmic_work(cfg, S, n_tiles)
        n_gemv_per_token = 4 * cfg.n_layers + 1
        emb = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles])[4]
        _, last = eng.prefill(emb, [S])
        eng.decode_begin(last.argmax(-1).to(torch.int32), [S], [S])
        eng.prof_enable(True, B.K_GEMV)
        n_prof = 16
        eng.decode_steps(n_prof)
        torch.cuda.synchronize()
        n_l, ms = eng.prof_read()
        eng.prof_enable(False)
        avg_ms = ms / max(1, n_l)
        bytes_per_launch = w_bytes / n_gemv_per_token
        ach = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "gemv_bf16_kernel<1> (decode weight stream)", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(),
                "launches": n_l, "avg_launch_ms": round(avg_ms, 5),
                "algorithmic_bytes_per_launch": int(bytes_per_launch),
                "method": f"HIP events around every GEMV launch of {n_prof} eager decode steps after the timed region"}
        eng.prof_enable(True, B.K_GEMM)
        eng.prefill(emb, [S])
        torch.cuda.synchronize()
        n_l, ms = eng.prof_read()
        eng.prof_enable(False)
        tf = pre_flops / (ms * 1e-3) / 1e12
        roof_prefill = {"bound": "mfma", "kernel": "gemm_bf16_kernel (Zephyr prefill QKV/O/gate-up/down)", "achieved": round(tf, 1),
                        "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / MFMA_BF16_PEAK_TF, 4),
                        "traffic": None, "launches": n_l, "total_ms": round(ms, 3),
                        "algorithmic_flops": pre_flops}
        # the whole image->first-token phase against the MFMA peak: SURVEY section 8(d) algorithmic work of this request -
        # CLIP 381.9 GFLOP/tile, Q-Former 1186.7 GFLOP/tile at L = 1888, Zephyr prefill 30.224 TFLOP at S = 2048 minus the
        # all-position lm_head generate() does not run (0.537 TFLOP) - over the measured latency (attention, norms, splice included)
        if cfg.n_layers == 32 and S == 2048 and n_tiles == 5:
            ft_flops = n_tiles * (381.9e9 + 1186.7e9) + 30.224e12 - 0.537e12
            ft_tf = ft_flops / (ttft_sum / args.steps) / 1e12
            roof_prefill["image_to_first_token"] = {"achieved": round(ft_tf, 1), "unit": "TFLOP/s", "frac": round(ft_tf / MFMA_BF16_PEAK_TF, 4),
                                                    "algorithmic_flops": ft_flops}

    # ---- extra leg (never `value`): the same request on the W8A16 engine of SURVEY config 5 - e4m3 weights with per-row
    # power-of-two scales streamed by the decode GEMV, bf16 activations, bf16 MFMA prefill on the dequantised weights.
    # Runs in a CHILD process with a time limit (as the tensor-parallel leg does): whatever happens there - an exception, a
    # stall while the second engine is built - the bf16 line of this process is printed. ----
    fp8_leg = None
    if world == 1 and not args.no_fp8_leg:
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--fp8-leg-only", "--steps", str(args.steps), "--layers", str(args.layers),
               "--new-tokens", str(args.new_tokens)]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                fp8_leg = json.loads(lines[-1])
            else:
                fp8_leg = {"value": None, "error": f"child rc={r.returncode}: {(r.stderr or '')[-300:]}"}
        except subprocess.TimeoutExpired:
            fp8_leg = {"value": None, "error": "W8A16 child timed out after 240 s"}

    # ---- N > 1: a guarded tensor-parallel leg beside the replica measurement.  Every rank starts a CHILD process that
    # runs this script in tp mode (its own rendezvous port), so a failure or hang inside the collectives cannot take the
    # replica numbers down with it; rank 0 attaches the child's result (or the reason it is missing). ----
    tp_leg = None
    if world > 1 and not tp_mode and os.environ.get("VZ_BENCH_TP_LEG", "1") != "0":
        import subprocess
        barrier()
        env = dict(os.environ)
        env["VZ_BENCH_PARALLELISM"] = "tp"
        env["VZ_BENCH_TP_LEG"] = "0"
        env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 23)
        # under torch.distributed.run the workers are CLIENTS of the agent's store (TORCHELASTIC_USE_AGENT_STORE=True); the child
        # group lives on its own port, so its rank 0 must host the store itself - otherwise every child waits for a server forever
        for k in [k for k in env if k.startswith("TORCHELASTIC_")]:
            env.pop(k)
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--steps", str(min(args.steps, 2)), "--warmup", "1",
               "--layers", str(args.layers), "--new-tokens", str(args.new_tokens), "--no-cpu-baseline", "--no-fp8-leg"]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=200)
            if rank == 0:
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    c = json.loads(lines[-1])
                    tp_leg = {k: c[k] for k in ("value", "unit", "image_to_first_token_ms", "ms_per_step", "scaling")}
                    tp_leg["parallelism"] = c["config"]["parallelism"]
                else:
                    tp_leg = {"value": None, "error": f"child rc={r.returncode}: {(r.stderr or '')[-400:]}"}
        except subprocess.TimeoutExpired:
            if rank == 0:
                tp_leg = {"value": None, "error": "tensor-parallel child timed out after 200 s"}
        barrier()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        try:
            cpu = cpu_baseline(cfg, S, n_tiles, n_new)
        except Exception as ex:   # the GPU numbers stand on their own; say why the CPU leg is missing
            cpu = {"value": None, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
                   "sample": f"failed: {type(ex).__name__}: {ex}"}
    n_dec_tokens = (n_new - 1) * args.steps * (1 if tp_mode else world)
    line = {
        "metric": "decode tokens/sec (image->first-token ms alongside), Zephyr-7B anyres 5-tile",
        "value": round(n_dec_tokens / dec_sum, 2) if world == 1 else round(n_dec_tokens / dec_sum, 2),
        "unit": "tokens/s",
        "image_to_first_token_ms": round(ttft_sum / args.steps * 1e3, 2),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "strong" if tp_mode else "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic (hash-generated weights seed 0, N(0,1) tiles, uniform ids)",
        "config": {"workload": f"configs[2]: 5 anyres tiles (4 crops + 1 global) + {n_ids}-id prompt -> S={S}, "
                               f"{n_new} greedy new tokens, batch 1 per GPU",
                   "layers": cfg.n_layers, "seq_len": S, "new_tokens": n_new,
                   "parallelism": "single GPU" if world == 1 else (f"tp{world} (one request, Zephyr tensor-parallel over RCCL)" if tp_mode
                                                                          else f"dp{world} replicas (one request per GPU, no collective)")},
        "roofline": roof, "roofline_prefill": roof_prefill, "cpu_baseline": cpu,
    }
    if tune:
        line["config"]["tune"] = tune
    if tp_leg is not None:
        line["tensor_parallel"] = tp_leg
    if fp8_leg is not None:
        line["fp8_weights"] = fp8_leg
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
