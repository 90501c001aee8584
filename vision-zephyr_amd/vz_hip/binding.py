"""ctypes binding of libviszephyr_hip.so (include/viszephyr.h).

The product path has no CPU fallback: if the shared library is missing, or a call fails, this
module raises.  torch is used only for device memory and streams (tensor.data_ptr(),
torch.cuda.current_stream()).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libviszephyr_hip.so")

VZ_OK, VZ_ERR_ARG, VZ_ERR_HIP, VZ_ERR_STATE, VZ_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
VZ_ASYNC_STREAMK = 2
VZ_ASYNC_PERSIST = 3
VZ_ASYNC_ONESHOT = 4
VZ_ASYNC_ATTN_O = 5
ABI_VERSION = 10
ACT_NONE, ACT_QUICK_GELU, ACT_GELU_ERF, ACT_SWIGLU = 0, 1, 2, 3
K_GEMM, K_GEMV, K_ATTN, K_ATTN_DEC, K_NORM, K_OTHER, K_FUSED, K_COMM = range(8)


class VzConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int), ("inter", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int),
        ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("vocab", C.c_int),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("sliding_window", C.c_int),
        ("clip_hidden", C.c_int), ("clip_inter", C.c_int), ("clip_layers", C.c_int), ("clip_heads", C.c_int),
        ("clip_image", C.c_int), ("clip_patch", C.c_int), ("clip_eps", C.c_float),
        ("qf_queries", C.c_int), ("qf_blocks", C.c_int), ("qf_heads", C.c_int), ("qf_kv_dim", C.c_int),
        ("qf_eps", C.c_float),
        ("fusion_groups", C.c_int), ("fusion_layers_per_group", C.c_int),
        ("max_batch", C.c_int), ("max_ctx", C.c_int), ("max_tiles", C.c_int), ("max_text", C.c_int),
        ("tp_size", C.c_int), ("tp_rank", C.c_int), ("clip_keep_cls", C.c_int), ("weight_fp8", C.c_int),
    ]


# every symbol include/viszephyr.h declares: (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_long, C.c_float
SYMBOLS = {
    "vz_last_error": (C.c_char_p, []),
    "vz_abi_version": (_I, []),
    "vz_target_arch": (C.c_char_p, []),
    "vz_op_linear": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "vz_op_tile_weights": (_I, [_P, _I, _I, _I, _P, _P]),
    "vz_op_linear_tiled": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _F, _P]),
    "vz_op_tile_weights_fp8": (_I, [_P, _I, _I, _I, _P, _P]),
    "vz_op_linear_tiled_fp8": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "vz_engine_prefill_fp8": (_I, [_P, _I]),
    "vz_op_rmsnorm_quant_fp8": (_I, [_P, _I, _P, _F, _P, _I, _P, _I, _I, _P]),
    "vz_op_quant_rows_fp8": (_I, [_P, _I, _P, _I, _P, _I, _I, _P]),
    "vz_op_linear_fp8_mfma": (_I, [_P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "vz_op_linear_impl": (_I, [_I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P]),
    "vz_op_causal_lm_loss": (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    "vz_op_layernorm": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _F, _P]),
    "vz_op_rmsnorm": (_I, [_P, _I, _P, _I, _P, _I, _I, _F, _P]),
    "vz_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I] + [_L] * 12 + [_F, _I, _I, _I, _P, _P]),
    "vz_op_transpose": (_I, [_P, _L, _P, _L, _I, _I, _P]),
    "vz_op_colsum": (_I, [_P, _I, _L, _I, _P, _L, _P, _P]),
    "vz_op_colsum_groups": (_I, [_L]),
    "vz_op_attention_bwd": (_I, [_P] * 7 + [_I] * 7 + [_F, _I, _I, _P, _P, _L, _P]),
    "vz_op_attention_split": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I] + [_L] * 12 + [_F, _I, _I, _I, _P, _P, _L, _P]),
    "vz_op_argmax": (_I, [_P, _I, _I, _P, _P]),
    "vz_op_rope_kv": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vz_op_attention_decode": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P]),
    "vz_op_attention_decode_fused": (_I, [_P] * 10 + [_I] * 7 + [_F, _P]),
    "vz_engine_create": (_I, [C.POINTER(VzConfig), C.POINTER(_P)]),
    "vz_engine_destroy": (_I, [_P]),
    "vz_engine_set_weight": (_I, [_P, C.c_char_p, _P, _I, _L]),
    "vz_engine_finalize": (_I, [_P]),
    "vz_engine_set_rope": (_I, [_P, _P, _P, _I]),
    "vz_comm_unique_id": (_I, [C.c_char_p]),
    "vz_comm_init": (_I, [_P, C.c_char_p]),
    "vz_clip_fused_features": (_I, [_P, _P, _I, _P, _P, _P]),
    "vz_qformer": (_I, [_P, _P, _I, _P, _I, _I, C.POINTER(C.c_int), _P, _P]),
    "vz_embed_splice": (_I, [_P, _P, _P, _I, _P, _P, _P]),
    "vz_llm_prefill": (_I, [_P, _P, _I, _I, C.POINTER(C.c_int), _P, _P, _P, _P]),
    "vz_llm_prefill_rows": (_I, [_P, _I, _P, _I, _I, C.POINTER(C.c_int), _P, _P, _P, _P]),
    "vz_llm_decode_set_row": (_I, [_P, _I, _I, _I, _I, _P]),
    "vz_llm_kv_move_rows": (_I, [_P, _I, _P, _P, _P, _P]),
    "vz_llm_decode_begin": (_I, [_P, _I, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "vz_llm_decode_steps": (_I, [_P, _I, _P, _P, _P]),
    "vz_llm_decode_mode": (_I, [_P, _P, _P]),
    "vz_test_persist_poke": (_I, [_P, _I, C.c_uint, _P, _P]),
    "vz_prof_persist_stamps": (_I, [_P, _P, _I]),
    "vz_op_oneshot_area_bytes": (C.c_size_t, [_I, _I]),
    "vz_op_allreduce_oneshot": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P]),
    "vz_test_allreduce_oneshot_all": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _P]),
    "vz_comm_oneshot_local": (_I, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "vz_comm_oneshot_attach": (_I, [_P, _P, _I]),
    "vz_engine_async_error": (_I, [_P, _P]),
    "vz_tune_set": (_I, [_I, _I]),
    "vz_engine_resize_vocab": (_I, [_P, _I]),
    "vz_op_resample_u8": (_I, [_P, _I, _I, _P, _P, _I, _I, _P, _P, _I, _P, _P, _I, _P]),
    "vz_op_anyres_tiles": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "vz_op_linear_rmsnorm": (_I, [_P, _I, _P, _F, _P, _I, _P, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "vz_op_linear_fp8": (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _P, _F, _P]),
    "vz_tp_all_gather": (_I, [_P, _P, _P, C.c_size_t, _P]),
    "vz_prof_enable": (_I, [_P, _I, _I]),
    "vz_prof_read": (_I, [_P, C.POINTER(C.c_long), C.POINTER(C.c_double)]),
    "vz_prof_attn_stamps": (_I, [C.POINTER(C.c_longlong)]),
    "vz_prof_gemm_stamps": (_I, [C.POINTER(C.c_longlong), _I, C.POINTER(_I)]),
    "vz_op_async_error": (_I, [_P, C.POINTER(_I)]),
    "vz_test_corrupt_streamk": (_I, [_P, _I, _I, _I]),
    "vz_engine_unset_weight": (_I, [_P, C.c_char_p]),
    "vz_llm_decode_sampling": (_I, [_P, _I, _F, _I, _F, C.c_ulonglong, _I]),
    "vz_llm_decode_ring": (_I, [_P, _P, _I, _I]),
    "vz_train_create": (_I, [_P, C.POINTER(_P), _P]),
    "vz_train_destroy": (_I, [_P]),
    "vz_train_set_master": (_I, [_P, C.c_char_p, _P, _L, _P]),
    "vz_train_param_count": (_I, [_P]),
    "vz_train_param_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_L), C.POINTER(_I)]),
    "vz_train_arenas": (_I, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_L)]),
    "vz_train_stage1_accumulate": (_I, [_P, _P, _I, _P, _I, _I, C.POINTER(_I), _P, _P, _P, _I, _I, C.POINTER(_I), _P, _P, _F, _P]),
    "vz_train_loss_sum": (_I, [_P, C.POINTER(C.c_double), _P]),
    "vz_train_zero_grad": (_I, [_P, _P]),
    "vz_train_comm_init": (_I, [_P, C.c_char_p, _I, _I]),
    "vz_train_allreduce": (_I, [_P, _P]),
    "vz_train_adamw_step": (_I, [_P, _F, _F, _F, _F, _F, _P]),
    "vz_op_vip_point": (_I, [_P, _I, _I, _I, _I, _I, _I, C.c_uint, _P]),
    "vz_op_sample": (_I, [_P, _I, _I, _F, _I, _F, C.c_ulonglong, _I, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen the HIP library and type every entry point.  Raises if it is missing: there is no
    eager/CPU fallback for the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `make -C vision-zephyr_amd/csrc` (or __graft_entry__.build()). "
            "The Vision-Zephyr MI355X path has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib() -> C.CDLL:
    return load_library()


class VzError(RuntimeError):
    pass


def check(rc: int):
    if rc == VZ_OK:
        return
    msg = lib().vz_last_error().decode(errors="replace")
    if rc == VZ_ERR_ARG:
        raise ValueError(msg)
    if rc == VZ_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise VzError(f"libviszephyr_hip error {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("libviszephyr_hip operates on device tensors only (no CPU fallback)")


# ------------------------------------------------------------------------------------------------
# operator wrappers (used by tests and by nothing else in the product: the engine calls the same
# launchers natively)
# ------------------------------------------------------------------------------------------------
def linear(x: torch.Tensor, w: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE, out_fp32=False,
           impl: Optional[int] = None) -> torch.Tensor:
    """epi(x[M,K] @ w[N,K]^T); bf16 in, bf16 (or fp32) out."""
    _need_cuda(x, w, bias, residual)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2
    assert x.stride(1) == 1 and w.stride(1) == 1
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
    if bias is not None:
        assert bias.dtype == torch.float32
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.stride(1) == 1
    args = (ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias), ptr(residual),
            0 if residual is None else residual.stride(0), act, int(out_fp32), stream_ptr(x.device))
    if impl is None:
        check(lib().vz_op_linear(*args))
    else:
        check(lib().vz_op_linear_impl(impl, *args))
    return out


def tile_weights(w: torch.Tensor) -> torch.Tensor:
    """fragment-tiled copy of a dense bf16 [N, K] weight for the 2..64-row MFMA weight stream (layout: include/viszephyr.h)."""
    _need_cuda(w)
    assert w.dtype == torch.bfloat16 and w.dim() == 2 and w.stride(1) == 1
    wt = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
    check(lib().vz_op_tile_weights(ptr(w), w.shape[0], w.shape[1], w.stride(0), ptr(wt), stream_ptr(w.device)))
    return wt


def oneshot_area(n_ranks: int, max_elems: int, device) -> torch.Tensor:
    """zero-filled receive area of ONE rank of the one-shot all-reduce (comm_oneshot.hip)"""
    return torch.zeros(int(lib().vz_op_oneshot_area_bytes(n_ranks, max_elems)), dtype=torch.uint8, device=device)


def oneshot_seq(device) -> torch.Tensor:
    """the two device words {sequence number = 1, ticket = 0} of one rank"""
    return torch.tensor([1, 0], dtype=torch.int32, device=device)


def allreduce_oneshot(areas, rank: int, max_elems: int, x: torch.Tensor, out: torch.Tensor, seq: torch.Tensor, err: torch.Tensor, stream=None):
    """rank `rank`'s launch of one all-reduce over len(areas) ranks (areas[q] = rank q's receive area); on torch's current stream"""
    _need_cuda(x, out, seq, err, *areas)
    arr = (C.c_void_p * len(areas))(*[a.data_ptr() for a in areas])
    check(lib().vz_op_allreduce_oneshot(arr, rank, len(areas), max_elems, ptr(x), ptr(out), x.numel(), ptr(seq), ptr(err),
                                        stream_ptr(x.device) if stream is None else stream))


def allreduce_oneshot_all(areas, max_elems: int, xs, outs, seqs, err: torch.Tensor):
    """TEST FORM: every rank of one all-reduce as a slice of ONE launch (one process, one GPU)"""
    n = len(areas)
    _need_cuda(err, *areas, *xs, *outs, *seqs)
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])      # noqa: E731
    check(lib().vz_test_allreduce_oneshot_all(arr(areas), n, max_elems, arr(xs), arr(outs), xs[0].numel(), arr(seqs), ptr(err), stream_ptr(err.device)))


def tile_weights_fp8(w8: torch.Tensor) -> torch.Tensor:
    """fragment-tiled copy of dense e4m3 rows (uint8 [N, K]) for the 17..64-row W8A16 stream of gemm_wide.hip."""
    _need_cuda(w8)
    assert w8.dtype == torch.uint8 and w8.dim() == 2 and w8.stride(1) == 1
    wt = torch.empty(w8.numel(), dtype=torch.uint8, device=w8.device)
    check(lib().vz_op_tile_weights_fp8(ptr(w8), w8.shape[0], w8.shape[1], w8.stride(0), ptr(wt), stream_ptr(w8.device)))
    return wt


def linear_tiled_fp8(x: torch.Tensor, w8t: torch.Tensor, wscale: torch.Tensor, N: int, bias=None, residual=None, act: int = ACT_NONE,
                     out_fp32=False) -> torch.Tensor:
    """epi(x @ (wscale[:, None] * e4m3(W8))^T) for 17 <= M <= 64 rows on the tiled e4m3 copy (gemm_wide.hip, W8A16)."""
    _need_cuda(x, w8t, wscale, bias, residual)
    M, K = x.shape
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
    check(lib().vz_op_linear_tiled_fp8(ptr(x), x.stride(0), ptr(w8t), ptr(wscale), ptr(out), out.stride(0), M, N, K, ptr(bias),
                                       ptr(residual), 0 if residual is None else residual.stride(0), act, int(out_fp32), stream_ptr(x.device)))
    return out


def linear_tiled(x: torch.Tensor, w: torch.Tensor, wt: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE, out_fp32=False,
                 norm_w=None, norm_eps: float = 0.0) -> torch.Tensor:
    """the MFMA weight stream (2 <= M <= 64) reading the tiled copy `wt` of w; bit-identical to linear(impl=3)."""
    _need_cuda(x, w, wt, bias, residual, norm_w)
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
    check(lib().vz_op_linear_tiled(ptr(x), x.stride(0), ptr(w), ptr(wt), w.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias),
                                   ptr(residual), 0 if residual is None else residual.stride(0), act, int(out_fp32), ptr(norm_w),
                                   float(norm_eps), stream_ptr(x.device)))
    return out


def linear_rmsnorm(x: torch.Tensor, norm_w: torch.Tensor, eps: float, w: torch.Tensor, residual=None, act: int = ACT_NONE,
                   out_fp32=False) -> torch.Tensor:
    """epi(rmsnorm(x) @ w^T) with the norm fused into the weight-stream kernel's staging; M <= 16."""
    _need_cuda(x, w, norm_w, residual)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and norm_w.dtype == torch.float32
    M, K = x.shape
    N = w.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
    check(lib().vz_op_linear_rmsnorm(ptr(x), x.stride(0), ptr(norm_w), float(eps), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K,
                                     ptr(residual), 0 if residual is None else residual.stride(0), act, int(out_fp32), stream_ptr(x.device)))
    return out


def linear_fp8(x: torch.Tensor, w8: torch.Tensor, scale: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE,
               out_fp32=False, norm_w=None, norm_eps: float = 0.0) -> torch.Tensor:
    """W8A16 weight stream: epi(x[M<=64,K] @ (scale[:,None] * e4m3(w8))[N,K]^T); x bf16, w8 uint8 (e4m3 bytes), scale fp32 [N]."""
    _need_cuda(x, w8, scale, bias, residual, norm_w)
    assert x.dtype == torch.bfloat16 and w8.dtype == torch.uint8 and scale.dtype == torch.float32
    assert x.dim() == 2 and w8.dim() == 2 and x.stride(1) == 1 and w8.stride(1) == 1
    M, K = x.shape
    N = w8.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x.device)
    check(lib().vz_op_linear_fp8(ptr(x), x.stride(0), ptr(w8), w8.stride(0), ptr(scale), ptr(out), out.stride(0), M, N, K,
                                 ptr(bias), ptr(residual), 0 if residual is None else residual.stride(0), act, int(out_fp32),
                                 ptr(norm_w), float(norm_eps), stream_ptr(x.device)))
    return out


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor):
    """(loss, valid target count) of fp32 logits [B, S, V] against labels [B, S] (-100 ignored), shifted by one, mean over valid targets."""
    _need_cuda(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 3 and logits.is_contiguous()
    Bn, S, V = logits.shape
    lab = labels.to(logits.device, torch.int32).contiguous()
    assert tuple(lab.shape) == (Bn, S)
    rows = torch.empty(Bn * S, dtype=torch.float32, device=logits.device)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(lib().vz_op_causal_lm_loss(ptr(logits), Bn, S, V, ptr(lab), ptr(rows), ptr(out), stream_ptr(logits.device)))
    return out[0], out[1]


def quant_rows_fp8(x: torch.Tensor):
    """bf16 [M, K] -> (e4m3 bytes uint8 [M, K], fp32 [M] power-of-two row scales): the device twin of vz_hip.quant.quantize_rows."""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib().vz_op_quant_rows_fp8(ptr(x), x.stride(0), ptr(q), q.stride(0), ptr(sc), M, K, stream_ptr(x.device)))
    return q, sc


def rmsnorm_quant_fp8(x: torch.Tensor, w: torch.Tensor, eps: float):
    """rmsnorm(x) rounded to bf16, then quant_rows_fp8 - in one launch; returns (e4m3 bytes [M, K], fp32 [M] scales)."""
    _need_cuda(x, w)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, K = x.shape
    q = torch.empty(M, K, dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    check(lib().vz_op_rmsnorm_quant_fp8(ptr(x), x.stride(0), ptr(w), float(eps), ptr(q), q.stride(0), ptr(sc), M, K, stream_ptr(x.device)))
    return q, sc


def linear_fp8_mfma(x8: torch.Tensor, xs: torch.Tensor, w8: torch.Tensor, ws: torch.Tensor, bias=None, residual=None, act: int = ACT_NONE,
                    out_fp32=False) -> torch.Tensor:
    """epi((x8 . w8^T) * xs[:, None] * ws[None, :]) on the fp8 MFMA; x8 / w8 uint8 e4m3 bytes, xs / ws fp32 row scales."""
    _need_cuda(x8, xs, w8, ws, bias, residual)
    assert x8.dtype == torch.uint8 and w8.dtype == torch.uint8 and xs.dtype == torch.float32 and ws.dtype == torch.float32
    M, K = x8.shape
    N = w8.shape[0]
    n_out = N // 2 if act == ACT_SWIGLU else N
    out = torch.empty(M, n_out, dtype=torch.float32 if out_fp32 else torch.bfloat16, device=x8.device)
    check(lib().vz_op_linear_fp8_mfma(ptr(x8), x8.stride(0), ptr(xs), ptr(w8), w8.stride(0), ptr(ws), ptr(out), out.stride(0), M, N, K,
                                      ptr(bias), ptr(residual), 0 if residual is None else residual.stride(0), act, int(out_fp32),
                                      stream_ptr(x8.device)))
    return out


def layernorm(x, w, b, eps):
    _need_cuda(x, w, b)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    y = torch.empty_like(x)
    check(lib().vz_op_layernorm(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(w), ptr(b), x.shape[0], x.shape[1],
                                eps, stream_ptr(x.device)))
    return y


def rmsnorm(x, w, eps):
    _need_cuda(x, w)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    y = torch.empty_like(x)
    check(lib().vz_op_rmsnorm(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(w), x.shape[0], x.shape[1], eps,
                              stream_ptr(x.device)))
    return y


def transpose(x):
    """bf16 [R, C] (unit column stride) -> contiguous [C, R] (vz_op_transpose)"""
    _need_cuda(x)
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    y = torch.empty(x.shape[1], x.shape[0], dtype=torch.bfloat16, device=x.device)
    check(lib().vz_op_transpose(ptr(x), x.stride(0), ptr(y), y.stride(0), x.shape[0], x.shape[1], stream_ptr(x.device)))
    return y


def colsum(y, out):
    """out[c] += sum_r y[r][c]: y bf16 [rows, cols] (unit column stride), out fp32 [cols] (vz_op_colsum, the bias-gradient reduction)"""
    _need_cuda(y, out)
    assert y.dtype == torch.bfloat16 and y.stride(1) == 1 and out.dtype == torch.float32 and out.is_contiguous()
    rows, cols = y.shape
    part = torch.empty(lib().vz_op_colsum_groups(rows) * cols, dtype=torch.float32, device=y.device)
    check(lib().vz_op_colsum(ptr(y), y.stride(0), rows, cols, ptr(part), part.numel(), ptr(out), stream_ptr(y.device)))
    return out


def attention_bwd(q, k, v, dO, scale, causal=False, window=0, kv_len=None, dkv_fp32=True):
    """Tile-resident attention backward (vz_op_attention_bwd, head_dim 128): q, dO [B,Sq,Hq,D], k / v [B,Hkv,Sk,D], all contiguous bf16
    -> dq [B,Sq,Hq,D] bf16, dk, dv [B,Hkv,Sk,D] fp32 (or bf16), summed over the query heads of a KV head."""
    _need_cuda(q, k, v, dO, kv_len)
    B, Sq, Hq, D = q.shape
    Hkv, Sk = k.shape[1], k.shape[2]
    for t in (q, k, v, dO):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    assert dO.shape == q.shape and v.shape == k.shape
    if kv_len is not None:
        assert kv_len.dtype == torch.int32
    dq = torch.empty_like(q)
    dk = torch.empty(B, Hkv, Sk, D, dtype=torch.float32 if dkv_fp32 else torch.bfloat16, device=q.device)
    dv = torch.empty_like(dk)
    ws = torch.empty(2 * B * Hq * Sq + 64, dtype=torch.float32, device=q.device)
    check(lib().vz_op_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(dO), ptr(dq), ptr(dk), ptr(dv), int(dkv_fp32), B, Sq, Sk, Hq, Hkv, D,
                                    float(scale), int(causal), int(window), ptr(kv_len), ptr(ws), ws.numel(), stream_ptr(q.device)))
    return dq, dk, dv


def attention(q, k, v, scale, causal=False, q_pos0=0, window=0, kv_len=None, workspace=None):
    """q [B,Sq,Hq,D], k/v [B,Sk,Hkv,D] (any strides with unit stride on D) -> [B,Sq,Hq,D].
    workspace: fp32 tensor; lets a few-row head_dim-512 launch split its keys over workgroups (vz_op_attention_split)."""
    _need_cuda(q, k, v, kv_len)
    B, Sq, Hq, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    for t in (q, k, v):
        assert t.dtype == torch.bfloat16 and t.stride(3) == 1
    o = torch.empty(B, Sq, Hq, D, dtype=torch.bfloat16, device=q.device)
    if kv_len is not None:
        assert kv_len.dtype == torch.int32
    if workspace is not None:
        _need_cuda(workspace)
        assert workspace.dtype == torch.float32 and workspace.is_contiguous()
        check(lib().vz_op_attention_split(ptr(q), ptr(k), ptr(v), ptr(o), B, Sq, Sk, Hq, Hkv, D,
                                          q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                          v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
                                          float(scale), int(causal), q_pos0, window, ptr(kv_len), ptr(workspace),
                                          workspace.numel(), stream_ptr(q.device)))
        return o
    check(lib().vz_op_attention(ptr(q), ptr(k), ptr(v), ptr(o), B, Sq, Sk, Hq, Hkv, D,
                                q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
                                float(scale), int(causal), q_pos0, window, ptr(kv_len), stream_ptr(q.device)))
    return o


def argmax(logits):
    _need_cuda(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.is_contiguous()
    ids = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    check(lib().vz_op_argmax(ptr(logits), logits.shape[0], logits.shape[1], ptr(ids), stream_ptr(logits.device)))
    return ids


def sample(logits, temperature: float, top_k: int = 0, top_p: float = 1.0, seed: int = 0, counter: int = 0):
    """one draw per row of fp32 logits [rows, V] on the device (sampling.hip): int32 [rows]."""
    _need_cuda(logits)
    assert logits.dtype == torch.float32 and logits.dim() == 2 and logits.is_contiguous()
    ids = torch.empty(logits.shape[0], dtype=torch.int32, device=logits.device)
    check(lib().vz_op_sample(ptr(logits), logits.shape[0], logits.shape[1], float(temperature), int(top_k or 0),
                             float(1.0 if top_p is None else top_p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(counter), ptr(ids),
                             stream_ptr(logits.device)))
    return ids


def op_async_error(device=None) -> int:
    """async error word of op-level launches on the current stream (VZ_ASYNC_*; 0 = none); blocking, clears it."""
    err = C.c_int(0)
    check(lib().vz_op_async_error(stream_ptr(device), C.byref(err)))
    return err.value


def rope_kv(qkv, cos, sin, pos, slot, kcache, vcache, B_, S, Hq, Hkv, D):
    """qkv bf16 [B*S,(Hq+2Hkv)*D]; returns rotated q [B*S,Hq,D]; K/V written into the caches in place."""
    _need_cuda(qkv, cos, sin, pos, slot, kcache, vcache)
    q = torch.empty(B_ * S, Hq, D, dtype=torch.bfloat16, device=qkv.device)
    check(lib().vz_op_rope_kv(ptr(qkv), qkv.stride(0), ptr(q), ptr(kcache), ptr(vcache), ptr(cos), ptr(sin), ptr(pos),
                              ptr(slot), B_, S, Hq, Hkv, D, kcache.shape[2], stream_ptr(qkv.device)))
    return q


def attention_decode(q, kcache, vcache, ctx_len, scale, nsplit=8, window=0):
    """q bf16 [B,Hq,D]; caches [B,Hkv,max_ctx,D]; ctx_len int32 [B] -> [B,Hq,D]."""
    _need_cuda(q, kcache, vcache, ctx_len)
    Bn, Hq, D = q.shape
    ws = torch.empty(Bn * Hq * nsplit * (D + 2), dtype=torch.float32, device=q.device)
    o = torch.empty_like(q)
    check(lib().vz_op_attention_decode(ptr(q), ptr(kcache), ptr(vcache), ptr(o), ptr(ws), Bn, Hq, kcache.shape[1], D,
                                       kcache.shape[2], nsplit, window, float(scale), ptr(ctx_len), stream_ptr(q.device)))
    return o


def attention_decode_fused(qkv, kcache, vcache, cos, sin, pos, slot, Hq, scale, nsplit=32, window=0, ticket=None):
    """qkv bf16 [B,(Hq+2Hkv)*D] -> o [B,Hq,D]; appends the new K/V rows to the caches in place."""
    _need_cuda(qkv, kcache, vcache, cos, sin, pos, slot)
    Bn, Hkv, max_ctx, D = kcache.shape
    ws = torch.empty(Bn * Hkv * nsplit * (4 * D + 32), dtype=torch.float32, device=qkv.device)
    if ticket is None:
        ticket = torch.zeros(Bn * Hkv, dtype=torch.int32, device=qkv.device)
    o = torch.empty(Bn, Hq, D, dtype=torch.bfloat16, device=qkv.device)
    check(lib().vz_op_attention_decode_fused(ptr(qkv), ptr(kcache), ptr(vcache), ptr(o), ptr(ws), ptr(ticket), ptr(cos),
                                             ptr(sin), ptr(pos), ptr(slot), Bn, Hq, Hkv, D, max_ctx, nsplit, window,
                                             float(scale), stream_ptr(qkv.device)))
    return o, ticket
