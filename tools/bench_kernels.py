#!/usr/bin/env python
"""Kernel micro-benchmarks on the GPU box (one process, interleaved rounds, random data; cdna guide rule 24/25).

    python tools/bench_kernels.py gemv|gemm|attn|all

Weights cycle through a pool larger than the 256 MiB Infinity Cache so every launch streams from HBM,
as the real decode step does (14 GB of weights per token)."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
import torch  # noqa: E402

from vz_hip import binding as B  # noqa: E402

dev = "cuda"


def timed(fn, n_iter, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n_iter):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n_iter * 1e3   # us


def bench_gemv():
    shapes = [("qkv+norm", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate-up swiglu", 28672, 4096, 3), ("down", 4096, 14336, 0),
              ("lm_head fp32", 32000, 4096, 0)]
    res = {}
    for name, N, K, act in shapes:
        pool = max(2, int(1.2e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(pool)]
        x = torch.randn(1, K, device=dev).bfloat16()
        out32 = "fp32" in name
        rows = {}
        # prepared ctypes argument tuples: ~3 us of host time per launch, far below the kernels' 8-60 us
        n_out = N // 2 if act == 3 else N
        out = torch.empty(1, n_out, dtype=torch.float32 if out32 else torch.bfloat16, device=dev)
        st = B.stream_ptr()
        fn = B.lib().vz_op_linear_impl
        argl = [(1, B.ptr(x), K, B.ptr(w), K, B.ptr(out), n_out, 1, N, K, None, None, 0, act, int(out32), st) for w in ws]
        for rnd in range(3):
            for v in (1, 2, 3, 4, 5, 6):
                B.check(B.lib().vz_tune_set(0, v))
                us = timed(lambda i: fn(*argl[i % pool]), 6 * pool)
                rows.setdefault(v, []).append(us)
        B.check(B.lib().vz_tune_set(0, 0))
        mb = N * K * 2 / 1e6
        res[name] = {v: (min(t), mb / min(t) * 1e6 / 1e9) for v, t in rows.items()}
        print(f"gemv {name:16s} {mb:7.1f} MB: " + "  ".join(f"v{v}: {min(t):6.1f}us {mb / min(t):5.2f}TB/s" for v, t in rows.items()), flush=True)
        del ws
    return res


def bench_gemv_fp8():
    """W8A16 weight stream: e4m3 rows + per-row scale, same shapes as the bf16 decode GEMVs; bytes = N*K (1 per weight)."""
    from vz_hip import quant
    import ctypes as C
    shapes = [("qkv+norm", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate-up swiglu", 28672, 4096, 3), ("down", 4096, 14336, 0),
              ("lm_head fp32", 32000, 4096, 0)]
    for name, N, K, act in shapes:
        pool = max(2, int(0.8e9 // (N * K)))
        ws = []
        for _ in range(pool):
            w8, sc = quant.quantize_rows(torch.randn(N, K, device=dev) * 0.02)
            ws.append((w8, sc))
        x = torch.randn(1, K, device=dev).bfloat16()
        out32 = "fp32" in name
        n_out = N // 2 if act == 3 else N
        out = torch.empty(1, n_out, dtype=torch.float32 if out32 else torch.bfloat16, device=dev)
        st = B.stream_ptr()
        fn = B.lib().vz_op_linear_fp8
        argl = [(B.ptr(x), K, B.ptr(w8), K, B.ptr(sc), B.ptr(out), n_out, 1, N, K, None, None, 0, act, int(out32), None, C.c_float(0.0), st)
                for w8, sc in ws]
        rows = {}
        for rnd in range(3):
            for v in (0, 1):
                B.check(B.lib().vz_tune_set(0, v))
                rows.setdefault(v, []).append(timed(lambda i: fn(*argl[i % pool]), 6 * pool))
        B.check(B.lib().vz_tune_set(0, 0))
        mb = N * K / 1e6
        print(f"gemv fp8 {name:16s} {mb:7.1f} MB: " + "  ".join(f"{'R2U8' if v == 0 else 'R4U4'}: {min(t):6.1f}us {mb / min(t):5.2f}TB/s" for v, t in rows.items()), flush=True)
        del ws


def bench_preprocess():
    """anyres preprocessing of one 1920 x 804 frame (the VCR movie-still shape of SURVEY config 5): device vs the PIL + numpy host path."""
    import numpy as np
    from PIL import Image
    from vz_hip.preprocess import AnyresPreprocessor
    pins = [[336, 672], [672, 336], [336, 1008], [1008, 336], [672, 672]]
    rng = np.random.default_rng(0)
    for h, w in [(804, 1920), (480, 640)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        pre = AnyresPreprocessor("cuda:0")
        d_img = torch.from_numpy(img).cuda()
        pre(d_img, pins)
        us = min(timed(lambda i: pre(d_img, pins), 20) for _ in range(3))
        host_img = torch.from_numpy(img)
        t0 = time.perf_counter()
        for _ in range(5):
            pre(host_img, pins)
        torch.cuda.synchronize()
        us_h2d = (time.perf_counter() - t0) / 5 * 1e6
        try:
            from vis_zephyr.model.multi_scale_process import process_any_resolution_image
            from vis_zephyr.model.vision_encoder.vision_encoder import _make_image_processor
            proc = _make_image_processor("openai/clip-vit-large-patch14-336", 336)
            pil = Image.fromarray(img)
            process_any_resolution_image(pil, proc, pins)
            t0 = time.perf_counter()
            for _ in range(3):
                process_any_resolution_image(pil, proc, pins)
            cpu_us = (time.perf_counter() - t0) / 3 * 1e6
        except Exception as ex:      # noqa: BLE001
            cpu_us = float("nan")
            print("host path unavailable:", ex)
        n = len(pre(d_img, pins))
        print(f"preprocess {h}x{w} -> {n} tiles: device {us:7.1f} us (image resident), {us_h2d:8.1f} us from a host tensor (PCIe + sync); "
              f"host PIL + CLIPImageProcessor {cpu_us:9.1f} us  ({cpu_us / us:6.1f}x)", flush=True)


def bench_skinny():
    """2..16 activation rows through the decode linears: MFMA weight stream (gemm_skinny.hip) vs the GEMV (M <= 8) / tile GEMM."""
    shapes = [("qkv", 6144, 4096, 0), ("o", 4096, 4096, 0), ("gate-up swiglu", 28672, 4096, 3), ("down", 4096, 14336, 0)]
    for name, N, K, act in shapes:
        pool = max(2, int(1.0e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(pool)]
        row = []
        for M in (1, 2, 4, 8, 16):
            x = torch.randn(M, K, device=dev).bfloat16()
            us = {}
            for mode in (1, 0):
                B.check(B.lib().vz_tune_set(9, mode))
                us[mode] = min(timed(lambda i: B.linear(x, ws[i % pool], act=act), 4 * pool) for _ in range(3))
            B.check(B.lib().vz_tune_set(9, 1))
            row.append(f"M{M}: {us[1]:6.1f} (was {us[0]:6.1f}) us")
            if K == 4096 and name != "o":      # the decode step fuses the RMSNorm into these
                nw = torch.ones(K, device=dev)
                usn = min(timed(lambda i: B.linear_rmsnorm(x, nw, 1e-5, ws[i % pool], act=act), 4 * pool) for _ in range(3))
                row[-1] += f" [+norm {usn:6.1f}]"
        print(f"skinny {name:15s} {N * K * 2 / 1e6:6.1f} MB: " + "  ".join(row), flush=True)
        del ws


def bench_gemv_resident():
    """Does a GEMV run faster when its weights sit in the 256 MiB Infinity Cache (read by the previous kernel) than from
    HBM?  Same weights every launch (resident) vs a pool larger than the cache (streamed), per shape."""
    for name, N, K, act in [("o", 4096, 4096, 0), ("qkv", 6144, 4096, 0), ("down", 4096, 14336, 0), ("gate-up", 28672, 4096, 3)]:
        n_out = N // 2 if act == 3 else N
        x = torch.randn(1, K, device=dev).bfloat16()
        out = torch.empty(1, n_out, dtype=torch.bfloat16, device=dev)
        st = B.stream_ptr()
        fn = B.lib().vz_op_linear_impl
        pool = max(2, int(1.2e9 // (N * K * 2)))
        ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(pool)]
        argl = [(1, B.ptr(x), K, B.ptr(w), K, B.ptr(out), n_out, 1, N, K, None, None, 0, act, 0, st) for w in ws]
        mb = N * K * 2 / 1e6
        us_stream = min(timed(lambda i: fn(*argl[i % pool]), 4 * pool) for _ in range(3))
        us_res = min(timed(lambda i: fn(*argl[0]), 4 * pool) for _ in range(3))
        print(f"gemv {name:8s} {mb:6.1f} MB: streamed {us_stream:6.1f} us {mb / us_stream:5.2f} TB/s   resident {us_res:6.1f} us {mb / us_res:5.2f} TB/s", flush=True)
        del ws


def bench_gemm():
    shapes = [("llm qkv", 2048, 6144, 4096, 0), ("llm o", 2048, 4096, 4096, 0), ("llm gate-up", 2048, 28672, 4096, 3),
              ("llm down", 2048, 4096, 14336, 0), ("clip qkv", 2885, 3072, 1024, 0), ("clip fc1", 2885, 4096, 1024, 1),
              ("clip fc2", 2885, 1024, 4096, 0), ("qf ca_kv", 2880, 8192, 5120, 0), ("qf small", 160, 4096, 4096, 0),
              ("qf ffn1", 160, 8192, 4096, 2), ("qf blk0 kv", 1920, 8192, 4096, 0)]
    for name, M, N, K, act in shapes:
        x = torch.randn(M, K, device=dev).bfloat16()
        ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(3)]
        row = []
        if M <= 512:
            B.check(B.lib().vz_tune_set(3, 1))
            us = min(timed(lambda i: B.linear(x, ws[i % 3], act=act, impl=0), 12) for _ in range(3))
            B.check(B.lib().vz_tune_set(3, 0))
            row.append(f"128 no-splitK: {us:8.1f} us")
        for impl, sk in ((0, 0), (2, 0), (2, 2)):
            B.check(B.lib().vz_tune_set(4, sk))
            us = min(timed(lambda i: B.linear(x, ws[i % 3], act=act, impl=impl), 12) for _ in range(3))
            tf = 2.0 * M * N * K / us / 1e6
            row.append(f"{'128' if impl == 0 else ('256sk' if sk else '256')}: {us:8.1f} us {tf:7.1f} TF ({tf / 2500 * 100:4.1f}%)")
        B.check(B.lib().vz_tune_set(4, 1))
        us = min(timed(lambda i: B.linear(x, ws[i % 3], act=act), 12) for _ in range(3))
        row.append(f"dispatch: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF")
        B.check(B.lib().vz_tune_set(11, 1))          # A/B in one process: workgroups wait for their stores before ending
        us = min(timed(lambda i: B.linear(x, ws[i % 3], act=act), 12) for _ in range(3))
        B.check(B.lib().vz_tune_set(11, 0))
        row.append(f"drain-wait: {us:8.1f} us")
        print(f"gemm {name:12s} M{M} N{N} K{K}: " + "   ".join(row), flush=True)


def bench_gemm_square():
    """Full-round shapes: the guide quotes its 256^2 template at 4096^3 / 8192^3 on random operands."""
    shapes = [("4k^3", 4096, 4096, 4096, 0), ("8k^3", 8192, 8192, 8192, 0), ("4 rounds", 2048, 32768, 4096, 0),
              ("3.5 rounds", 2048, 28672, 4096, 0), ("3 rounds", 2048, 24576, 4096, 0), ("down 256", 2048, 4096, 14336, 0),
              ("s1 qkv", 12736, 6144, 4096, 0), ("s1 gate-up", 12736, 28672, 4096, 3), ("s1 down", 12736, 4096, 14336, 0), ("clip fc1 x64", 36928, 4096, 1024, 1)]
    for name, M, N, K, act in shapes:
        x = torch.rand(M, K, device=dev).bfloat16() * 2 - 1
        ws = [(torch.rand(N, K, device=dev) * 2 - 1).bfloat16() for _ in range(2)]
        row = []
        for impl, ring in ((0, 0), (2, 0), (2, 1)):      # 256 kernel without / with the stream-K tail
            B.check(B.lib().vz_tune_set(4, ring))
            us = min(timed(lambda i: B.linear(x, ws[i % 2], act=act, impl=impl), 10) for _ in range(3))
            tf = 2.0 * M * N * K / us / 1e6
            row.append(f"{'128' if impl == 0 else ('256sk' if ring else '256')}: {us:8.1f} us {tf:7.1f} TF")
        B.check(B.lib().vz_tune_set(4, 1))
        us = min(timed(lambda i: B.linear(x, ws[i % 2], act=act), 12) for _ in range(3))
        row.append(f"dispatch: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF")
        print(f"gemm {name:12s} M{M} N{N} K{K}: " + "   ".join(row), flush=True)
        del ws


def bench_attn():
    S, Hq, Hkv, D = 2048, 32, 8, 128
    q = torch.randn(1, S, Hq, D, device=dev).bfloat16()
    k = torch.randn(1, S, Hkv, D, device=dev).bfloat16()
    v = torch.randn(1, S, Hkv, D, device=dev).bfloat16()
    fl = 4.0 * S * S * D * Hq / 2
    res = {1: [], 2: [], 3: []}
    for rnd in range(3):
        for ver in (1, 2, 3):
            B.check(B.lib().vz_tune_set(2, ver))
            res[ver].append(timed(lambda i: B.attention(q, k, v, D ** -0.5, True, 0, 4096), 10))
    B.check(B.lib().vz_tune_set(2, 3))
    print(f"attn prefill causal S{S}: " + "  ".join(f"v{ver}: {min(t):7.1f} us {fl / min(t) / 1e6:6.1f} TF" for ver, t in res.items()), flush=True)
    kc = torch.randn(1, Hkv, S, D, device=dev).bfloat16()       # cache layout: keys of one head contiguous
    vc = torch.randn(1, Hkv, S, D, device=dev).bfloat16()
    us = min(timed(lambda i: B.attention(q, kc.permute(0, 2, 1, 3), vc.permute(0, 2, 1, 3), D ** -0.5, True, 0, 4096), 10) for _ in range(3))
    print(f"attn prefill causal S{S} (KV-cache layout): v2 {us:7.1f} us {fl / us / 1e6:6.1f} TF", flush=True)
    T = 5
    qkv = torch.randn(T, 577, 3 * 1024, device=dev).bfloat16()
    res = {1: [], 2: [], 3: []}
    for rnd in range(3):
        for ver in (1, 2, 3):
            B.check(B.lib().vz_tune_set(2, ver))
            res[ver].append(timed(lambda i: B.attention(qkv[:, :, :1024].view(T, 577, 16, 64), qkv[:, :, 1024:2048].view(T, 577, 16, 64),
                                                        qkv[:, :, 2048:].view(T, 577, 16, 64), 0.125), 10))
    B.check(B.lib().vz_tune_set(2, 3))
    cf = 4.0 * 577 * 577 * 64 * 16 * T
    print(f"attn clip T{T}: " + "  ".join(f"v{ver}: {min(t):7.1f} us {cf / min(t) / 1e6:6.1f} TF" for ver, t in res.items()), flush=True)
    # decode attention at ctx 2048: fused vs two-kernel
    max_ctx = 2304
    kc = torch.randn(1, Hkv, max_ctx, D, device=dev).bfloat16()
    vc = torch.randn(1, Hkv, max_ctx, D, device=dev).bfloat16()
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
    fr = torch.arange(max_ctx, dtype=torch.float32).unsqueeze(-1) * inv
    cos, sin = fr.cos().to(dev).contiguous(), fr.sin().to(dev).contiguous()
    qkv1 = torch.randn(1, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
    slot = torch.tensor([2047], dtype=torch.int32, device=dev)
    ticket = torch.zeros(Hkv, dtype=torch.int32, device=dev)
    for ns in (8, 16, 32, 64):
        us = min(timed(lambda i: B.attention_decode_fused(qkv1, kc, vc, cos, sin, slot, slot, Hq, D ** -0.5, ns, 4096, ticket), 50)
                 for _ in range(3))
        print(f"attn decode fused ctx2048 nsplit{ns}: {us:7.1f} us  ({2 * Hkv * 2048 * 256 / us / 1e3:6.1f} GB/s of KV)", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.manual_seed(0)
    for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):       # experiments: "knob=value,..." for vz_tune_set
        B.check(B.lib().vz_tune_set(*(int(v) for v in kv.split("="))))
    if what in ("gemv", "all"):
        bench_gemv()
    if what in ("gemm", "all"):
        bench_gemm()
    if what == "preprocess":
        bench_preprocess()
    if what == "skinny":
        bench_skinny()
    if what == "gemv8":
        bench_gemv_fp8()
    if what == "gemvres":
        bench_gemv_resident()
    if what == "gemmsq":
        bench_gemm_square()
    if what in ("attn", "all"):
        bench_attn()
