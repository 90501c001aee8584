// libviszephyr_hip.so - C ABI (include/viszephyr.h) and the stage engine of the Vision-Zephyr
// forward/generate path on gfx950.  The engine owns workspace + KV cache and sequences the
// hand-written kernels of one stage on the caller's HIP stream:
//
//   vz_clip_fused_features  a8-a10  ref:vis_zephyr/model/vision_encoder/vision_encoder.py:58-117
//   vz_qformer              a11     ref:vis_zephyr/model/multimodal_projector/builder.py:34-92
//   vz_embed_splice         a6/a7   ref:vis_zephyr/model/vis_zephyr_arch.py:236-305,476-530
//   vz_llm_prefill          a12     hf:models/mistral/modeling_mistral.py:202-241,340-466
//   vz_llm_decode_*         a13     hf:generation/utils.py greedy branch, one hipGraph replay per token
//
// Q-Former block 0 (ref builder.py:80-87) keeps only the first 32 rows of a block that ran on
// [32 queries ; L text tokens].  Those rows depend on the other rows only through the self-attention
// keys/values, and both the query rows and the text rows are identical for every tile of a sample,
// so the engine computes block 0's self-attention once per SAMPLE for the 32 query rows (K/V over
// all 32+L rows) and runs cross-attention/FFN per tile on 32 rows - identical results, without the
// rows the reference computes and throws away.
#include <algorithm>
#include <stdarg.h>
#include <stdlib.h>

#include <string>
#include <unordered_map>
#include <vector>

#include <rccl/rccl.h>

#include "vz_common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
thread_local hipEvent_t g_vz_prof_start = nullptr, g_vz_prof_stop = nullptr;
static thread_local char g_err[1024] = "";
void vz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vz_last_error(void) { return g_err; }
extern "C" int vz_abi_version(void) { return 10; }
extern "C" const char* vz_target_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------------------------------------
// operator entry points
// ------------------------------------------------------------------------------------------------
static LinearArgs mk_linear(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                            const float* bias, const void* residual, int ldr, int act, int out_fp32) {
    LinearArgs a;
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.bias = bias; a.residual = (const bf16_t*)residual; a.ldr = ldr;
    a.act = act; a.out_fp32 = out_fp32; a.norm_w = nullptr; a.norm_eps = 0.f; a.err = nullptr;
    return a;
}

// vz_tune_set(14, rows): from this many rows on a decode step's linears run on the 128^2 tile GEMM (65 = never)
static int g_decode_tile_rows = 29;      // measured cross-over on the tiled weight copies (profiles/r02_rows.txt): MFMA weight stream 4.42 ms per step at 25 rows, 4.74 at 32; tile route 4.56 / 4.64
static int g_fp8_prefill_min_rows = 768;   // vz_tune_set(22, rows): fewest prefill rows that take the fp8 MFMA path of a prefill_fp8 engine
static int g_decode_sk_short = 8;      // vz_tune_set(15, v): split-K factor of the K = 4096 decode projections (QKV, O) on the tile-GEMM route
static int vz_decode_splitk(int N, int K, int act) {
    // In situ (rocprofv3 of a 64-row step, profiles/r02_rows.txt) the 128^2 kernel is bound by the bytes its workgroups keep in flight
    // (32 KiB each): gate|up with 224 workgroups ran at 3.3 TB/s, down with 512 (split 16) at 4.4.  So every projection is cut along K
    // until ~2 workgroups per CU are streaming: gate|up x2 (the SwiGLU pairs are formed by the finalize kernel), QKV / O x8, down x16.
    if (N & 7) return 0;
    const int tiles_n = (N + 127) / 128;
    if (act == VZ_ACT_SWIGLU) return tiles_n < 384 ? 2 : 0;
    if (K < 8192) return g_decode_sk_short;
    int sk = (512 + tiles_n - 1) / tiles_n;
    if (sk > 16) sk = 16;
    return sk < 1 ? 1 : sk;
}

extern "C" int vz_op_linear(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K,
                            const float* bias, const void* residual, int ldr, int act, int out_fp32, vz_stream s) {
    return vz_launch_linear(mk_linear(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, act, out_fp32), (hipStream_t)s);
}
extern "C" int vz_op_linear_fp8(const void* A, int lda, const void* W8, int ldw, const float* wscale, void* C, int ldc, int M, int N,
                                int K, const float* bias, const void* residual, int ldr, int act, int out_fp32,
                                const float* norm_w, float norm_eps, vz_stream s) {
    VZ_CHECK_ARG(W8 && wscale, "linear_fp8: null weights / scales");
    LinearArgs a = mk_linear(A, lda, (const void*)W8, ldw, C, ldc, M, N, K, bias, residual, ldr, act, out_fp32);
    a.W8 = (const unsigned char*)W8; a.wscale = wscale; a.norm_w = norm_w; a.norm_eps = norm_eps;
    return vz_launch_linear(a, (hipStream_t)s);      // 1 row: GEMV; 2..16 rows: MFMA weight stream (gemm_skinny.hip)
}
extern "C" int vz_op_rmsnorm_quant_fp8(const void* x, int ldx, const float* w, float eps, void* q, int ldq, float* scale, int rows, int cols, vz_stream s) {
    return vz_launch_rmsnorm_quant_fp8((const bf16_t*)x, ldx, w, eps, (unsigned char*)q, ldq, scale, rows, cols, (hipStream_t)s);
}
extern "C" int vz_op_quant_rows_fp8(const void* x, int ldx, void* q, int ldq, float* scale, int rows, int K, vz_stream s) {
    return vz_launch_quant_rows_fp8((const bf16_t*)x, ldx, (unsigned char*)q, ldq, scale, rows, K, (hipStream_t)s);
}
extern "C" int vz_op_linear_fp8_mfma(const void* A8, int lda, const float* ascale, const void* W8, int ldw, const float* wscale, void* C, int ldc,
                                     int M, int N, int K, const float* bias, const void* residual, int ldr, int act, int out_fp32, vz_stream s) {
    Fp8LinearArgs a;
    a.A8 = (const unsigned char*)A8; a.lda = lda; a.ascale = ascale; a.W8 = (const unsigned char*)W8; a.ldw = ldw; a.wscale = wscale;
    a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias; a.residual = (const bf16_t*)residual; a.ldr = ldr; a.act = act; a.out_fp32 = out_fp32;
    return vz_launch_gemm_fp8(a, (hipStream_t)s);
}
extern "C" int vz_op_linear_rmsnorm(const void* A, int lda, const float* norm_w, float norm_eps, const void* W, int ldw, void* C, int ldc,
                                    int M, int N, int K, const void* residual, int ldr, int act, int out_fp32, vz_stream s) {
    VZ_CHECK_ARG(norm_w && M >= 1 && M <= 16, "linear_rmsnorm: needs norm weights and 1 <= M <= 16 (the fused norm lives on the <= 16-row weight-stream kernels)");
    LinearArgs a = mk_linear(A, lda, W, ldw, C, ldc, M, N, K, nullptr, residual, ldr, act, out_fp32);
    a.norm_w = norm_w; a.norm_eps = norm_eps;
    VZ_CHECK_ARG((g_skinny_mode && vz_skinny_ok(a)) || vz_gemv_ok(a), "linear_rmsnorm: shape M=%d K=%d not supported by the weight-stream kernels", M, K);
    return vz_launch_linear(a, (hipStream_t)s);
}
extern "C" int vz_op_tile_weights(const void* W, int N, int K, int ldw, void* Wt, vz_stream s) {
    return vz_launch_tile_weights((const bf16_t*)W, N, K, ldw, (bf16_t*)Wt, (hipStream_t)s);
}
extern "C" int vz_op_linear_tiled(const void* A, int lda, const void* W, const void* Wt, int ldw, void* C, int ldc, int M, int N, int K,
                                  const float* bias, const void* residual, int ldr, int act, int out_fp32, const float* norm_w,
                                  float norm_eps, vz_stream s) {
    VZ_CHECK_ARG(Wt && ldw == K && (N & 15) == 0, "linear_tiled: needs the tiled copy of a dense [N, K] weight with N %% 16 == 0");
    LinearArgs a = mk_linear(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, act, out_fp32);
    a.Wt = (const bf16_t*)Wt; a.norm_w = norm_w; a.norm_eps = norm_eps;
    if (M >= 17 && vz_wide_ok(a)) return vz_launch_wide(a, (hipStream_t)s);          // 17..64 rows: gemm_wide.hip
    VZ_CHECK_ARG(g_skinny_mode && vz_skinny_ok(a), "linear_tiled: the MFMA weight stream takes 2 <= M <= 64 (fused norm: <= 16), K %% 64 == 0 (M=%d K=%d)", M, K);
    return vz_launch_skinny(a, (hipStream_t)s);
}
extern "C" int vz_op_tile_weights_fp8(const void* W8, int N, int K, int ldw, void* W8t, vz_stream s) {
    return vz_launch_tile_weights_fp8((const unsigned char*)W8, N, K, ldw, (unsigned char*)W8t, (hipStream_t)s);
}
// 17..64 rows on the e4m3 fragment-tiled copy (gemm_wide.hip's W8A16 stream): C = epi((A . dequant(W8)^T)), bf16 activations
extern "C" int vz_op_linear_tiled_fp8(const void* A, int lda, const void* W8t, const float* wscale, void* C, int ldc, int M, int N, int K,
                                      const float* bias, const void* residual, int ldr, int act, int out_fp32, vz_stream s) {
    VZ_CHECK_ARG(W8t && wscale && (N & 127) == 0, "linear_tiled_fp8: needs the tiled e4m3 copy + row scales of a dense [N, K] weight with N %% 128 == 0");
    LinearArgs a = mk_linear(A, lda, W8t, K, C, ldc, M, N, K, bias, residual, ldr, act, out_fp32);      // (W is unused on this route: any aligned pointer)
    a.W8t = (const unsigned char*)W8t; a.wscale = wscale;
    VZ_CHECK_ARG(vz_wide_ok(a), "linear_tiled_fp8: needs 17 <= M <= 64, K %% 1024 == 0 (M=%d N=%d K=%d)", M, N, K);
    return vz_launch_wide(a, (hipStream_t)s);
}
extern "C" int vz_op_linear_impl(int impl, const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N,
                                 int K, const float* bias, const void* residual, int ldr, int act, int out_fp32,
                                 vz_stream s) {
    LinearArgs a = mk_linear(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, act, out_fp32);
    if (impl == 0) return vz_launch_gemm128(a, (hipStream_t)s);
    if (impl == 1) return vz_launch_gemv(a, (hipStream_t)s);
    if (impl == 2) return vz_launch_gemm256(a, (hipStream_t)s);
    if (impl == 3) return vz_launch_skinny(a, (hipStream_t)s);
    if (impl == 4) { a.splitk_hint = vz_decode_splitk(N, K, act); return vz_launch_gemm128(a, (hipStream_t)s); }     // the 17..64-row decode route
    vz_set_error("linear: unknown impl %d", impl);
    return VZ_ERR_ARG;
}
extern "C" int vz_op_causal_lm_loss(const float* logits, int B, int S, int V, const int* labels, float* loss_rows, float* out, vz_stream s) {
    return vz_launch_causal_lm_loss(logits, B, S, V, labels, loss_rows, out, (hipStream_t)s);
}
extern "C" int vz_op_layernorm(const void* x, int ldx, void* y, int ldy, const float* w, const float* b, int rows, int cols,
                               float eps, vz_stream s) {
    return vz_launch_layernorm((const bf16_t*)x, ldx, (bf16_t*)y, ldy, w, b, rows, cols, eps, (hipStream_t)s);
}
extern "C" int vz_op_rmsnorm(const void* x, int ldx, void* y, int ldy, const float* w, int rows, int cols, float eps,
                             vz_stream s) {
    return vz_launch_rmsnorm((const bf16_t*)x, ldx, (bf16_t*)y, ldy, w, rows, cols, eps, (hipStream_t)s);
}
extern "C" int vz_op_attention(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int Hq, int Hkv,
                               int head_dim, long q_bs, long q_ss, long q_hs, long k_bs, long k_ss, long k_hs, long v_bs,
                               long v_ss, long v_hs, long o_bs, long o_ss, long o_hs, float scale, int causal, int q_pos0,
                               int window, const int* kv_len, vz_stream s) {
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.head_dim = head_dim;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = q_hs; a.k_bs = k_bs; a.k_ss = k_ss; a.k_hs = k_hs;
    a.v_bs = v_bs; a.v_ss = v_ss; a.v_hs = v_hs; a.o_bs = o_bs; a.o_ss = o_ss; a.o_hs = o_hs;
    a.scale = scale; a.causal = causal; a.q_pos0 = q_pos0; a.window = window; a.kv_len = kv_len;
    return vz_launch_attention(a, (hipStream_t)s);
}
extern "C" int vz_op_attention_split(const void* q, const void* k, const void* v, void* o, int B, int Sq, int Sk, int Hq, int Hkv,
                                     int head_dim, long q_bs, long q_ss, long q_hs, long k_bs, long k_ss, long k_hs, long v_bs,
                                     long v_ss, long v_hs, long o_bs, long o_ss, long o_hs, float scale, int causal, int q_pos0,
                                     int window, const int* kv_len, float* ws, long ws_floats, vz_stream s) {
    VZ_CHECK_ARG(ws_floats >= 0 && (ws || ws_floats == 0) && ((uintptr_t)ws & 15) == 0, "attention_split: bad workspace");
    AttnArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)o;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = Hq; a.Hkv = Hkv; a.head_dim = head_dim;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = q_hs; a.k_bs = k_bs; a.k_ss = k_ss; a.k_hs = k_hs;
    a.v_bs = v_bs; a.v_ss = v_ss; a.v_hs = v_hs; a.o_bs = o_bs; a.o_ss = o_ss; a.o_hs = o_hs;
    a.scale = scale; a.causal = causal; a.q_pos0 = q_pos0; a.window = window; a.kv_len = kv_len;
    a.part = ws; a.part_floats = (size_t)ws_floats;
    return vz_launch_attention(a, (hipStream_t)s);
}
// two helpers of the training step at op level (parity tests): dst[c][r] = src[r][c], and out[c] += sum_r y[r][c] through `part`
extern "C" int vz_op_transpose(const void* src, long src_ld, void* dst, long dst_ld, int R, int C, vz_stream s) {
    return vz_launch_transpose((const bf16_t*)src, src_ld, 0, 0, (bf16_t*)dst, dst_ld, 0, 0, R, C, 1, 1, 0, (hipStream_t)s);
}
extern "C" int vz_op_colsum(const void* y, int ld, long rows, int cols, float* part, long part_floats, float* out, vz_stream s) {
    VZ_CHECK_ARG(part_floats >= (long)vz_colsum_groups(rows) * cols, "colsum: scratch of %ld floats, need %ld", part_floats, (long)vz_colsum_groups(rows) * cols);
    return vz_launch_colsum((const bf16_t*)y, ld, rows, cols, part, out, (hipStream_t)s);
}
extern "C" int vz_op_colsum_groups(long rows) { return vz_colsum_groups(rows); }
extern "C" int vz_op_attention_bwd(const void* q, const void* k, const void* v, const void* dO, void* dq, void* dk, void* dv, int dkv_fp32, int B,
                                   int Sq, int Sk, int Hq, int Hkv, int head_dim, float scale, int causal, int window, const int* kv_len,
                                   float* ws, long ws_floats, vz_stream s) {
    VZ_CHECK_ARG(B > 0 && Sq > 0 && Sk > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_bwd: bad shape");
    VZ_CHECK_ARG(head_dim == 128, "attention_bwd: the tile-resident backward is built for head_dim 128 (the Zephyr layers), got %d", head_dim);
    FlashBwdArgs f;
    f.q = (const bf16_t*)q; f.k = (const bf16_t*)k; f.v = (const bf16_t*)v; f.dO = (const bf16_t*)dO; f.B = B; f.Sq = Sq; f.Sk = Sk; f.Hq = Hq; f.Hkv = Hkv; f.D = head_dim;
    const long A = (long)Hq * head_dim;
    f.q_bs = f.o_bs = f.dq_bs = (long)Sq * A; f.q_ss = f.o_ss = f.dq_ss = A; f.q_hs = f.o_hs = f.dq_hs = head_dim;
    f.k_bs = f.v_bs = f.dk_bs = (long)Hkv * Sk * head_dim; f.k_ss = f.v_ss = f.dk_ss = head_dim; f.k_hs = f.v_hs = f.dk_hs = (long)Sk * head_dim;
    f.scale = scale; f.causal = causal; f.window = window; f.kv_len = kv_len;
    f.dq = (bf16_t*)dq; f.dk = dk; f.dv = dv; f.dkv_fp32 = dkv_fp32;
    return vz_launch_flash_bwd(f, ws, ws_floats > 0 ? (size_t)ws_floats * 4 : 0, (hipStream_t)s);
}
extern "C" int vz_op_rope_kv(const void* qkv, int ld, void* q_out, void* kc, void* vc, const float* cosT, const float* sinT,
                             const int* pos, const int* slot, int B, int S, int Hq, int Hkv, int D, int max_ctx, vz_stream s) {
    VZ_CHECK_ARG(qkv && q_out && kc && vc && cosT && sinT && pos && slot && B > 0 && S > 0, "rope: bad argument");
    return vz_launch_rope_kv((const bf16_t*)qkv, ld, (bf16_t*)q_out, (bf16_t*)kc, (bf16_t*)vc, cosT, sinT, pos, slot, B, S, Hq,
                             Hkv, D, max_ctx, (hipStream_t)s);
}
extern "C" int vz_op_attention_decode(const void* q, const void* kc, const void* vc, void* o, float* ws, int B, int Hq, int Hkv,
                                      int D, int max_ctx, int nsplit, int window, float scale, const int* ctx_len, vz_stream s) {
    VZ_CHECK_ARG(q && kc && vc && o && ws && ctx_len && B > 0, "attention_decode: bad argument");
    AttnDecodeArgs a;
    a.q = (const bf16_t*)q; a.kc = (const bf16_t*)kc; a.vc = (const bf16_t*)vc; a.o = (bf16_t*)o; a.part = ws;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.max_ctx = max_ctx; a.nsplit = nsplit; a.window = window; a.scale = scale;
    a.ctx_len = ctx_len;
    return vz_launch_attn_decode(a, (hipStream_t)s);
}
extern "C" int vz_op_attention_decode_fused(const void* qkv, void* kc, void* vc, void* o, float* ws, unsigned* ticket,
                                            const float* cosT, const float* sinT, const int* pos, const int* slot, int B, int Hq,
                                            int Hkv, int D, int max_ctx, int nsplit, int window, float scale, vz_stream s) {
    VZ_CHECK_ARG(qkv && kc && vc && o && ws && ticket && cosT && sinT && pos && slot && B > 0, "attention_decode_fused: bad argument");
    AttnDecodeFusedArgs a;
    a.qkv = (const bf16_t*)qkv; a.kc = (bf16_t*)kc; a.vc = (bf16_t*)vc; a.o = (bf16_t*)o; a.part = ws; a.ticket = ticket;
    a.cosT = cosT; a.sinT = sinT; a.pos = pos; a.slot = slot;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.max_ctx = max_ctx; a.nsplit = nsplit; a.window = window; a.scale = scale;
    return vz_launch_attn_decode_fused(a, (hipStream_t)s);
}
extern "C" int vz_op_argmax(const float* logits, int rows, int cols, int* ids, vz_stream s) {
    return vz_launch_argmax(logits, rows, cols, ids, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, (hipStream_t)s);
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
struct Weight { const void* p; int dtype; long n; };

enum { K_GEMM = 0, K_GEMV = 1, K_ATTN = 2, K_ATTN_DEC = 3, K_NORM = 4, K_OTHER = 5, K_FUSED = 6, K_COMM = 7 };

struct vz_engine {
    vz_config c;
    // tensor parallelism: this rank's share of the 32 query heads / 8 KV heads / 14336 MLP columns and
    // of the vocabulary (Vp = ceil(vocab / tp) rows of lm_head, zero padded); tp == 1: everything.
    int tp = 1, rank = 0, Hq_l = 0, Hkv_l = 0, I_l = 0, Vp = 0;
    ncclComm_t comm = nullptr;
    // one-shot all-reduce of the decode step (comm_oneshot.hip): this rank's receive area, every rank's area (peer-mapped), sequence words
    void* os_area = nullptr; void* os_areas[8] = {nullptr}; int os_ranks = 0; unsigned* os_seq = nullptr;
    static constexpr int OS_MAX_ELEMS = 64 * 4096;       // up to 64 decode rows of hidden 4096
    float* d_gather = nullptr;     // [tp][rows][Vp] all-gathered logits before the repack
    size_t gather_floats = 0;
    std::unordered_map<std::string, Weight> w;
    // row-major bf16 weight -> its fragment-tiled copy (registered as "<name>t", same element count): the 2..64-row decode
    // linears stream the tiled copy (gemm_skinny.hip); rebuilt whenever the registry changes
    std::unordered_map<const void*, const bf16_t*> tiled; bool tiled_dirty = true;
    std::unordered_map<const void*, const unsigned char*> tiled8;      // e4m3 copy "<name>8" / "<name>.w8" -> its fragment-tiled copy "<..>t" (dtype 2)
    bool finalized = false;
    // rope
    const float* cosT = nullptr; const float* sinT = nullptr; int rope_max = 0;
    // workspace (one arena, carved per stage; stages never overlap in time on a stream)
    char* arena = nullptr; size_t arena_bytes = 0;
    // kv cache: [layer][2][B][Hkv][max_ctx][D]
    bf16_t* kv = nullptr; size_t kv_layer_elems = 0;
    // decode state (device)
    int* d_state = nullptr;  // [cur_ids[B] | pos[B] | slot[B] | len[B] | step]
    int dec_B = 0;
    int prefill_fp8 = 0;         // vz_engine_prefill_fp8: the Zephyr prefill linears run e4m3 x e4m3 on the scaled MFMA (weight_fp8 engines)
    bool comm_graph_ok = true;   // RCCL collectives captured into the decode graph (cleared if a capture is refused -> eager steps)
    int dec_len_max = 0;         // host-side bound on the longest row's visible keys (grows by one per launched step)
    // host mirror of the device-side decode state, per row: keys visible to the NEXT step, its rotary position, and whether the
    // row is parked (continuous batching: ctx_len 0, steps harmlessly, never checked against the capacity)
    std::vector<int> h_len, h_pos; std::vector<char> h_parked;
    // sampling tail (vz_llm_decode_sampling): off = greedy argmax
    int samp_on = 0, samp_top_k = 0, samp_ctr0 = 0; float samp_temp = 1.f, samp_top_p = 1.f; unsigned samp_seed[2] = {0, 0};
    int* ring = nullptr; int ring_n = 0, ring_rows = 0;   // host-visible token ring of the streamer path (vz_llm_decode_ring): [ring_rows][ring_n]
    hipStream_t last_stream = nullptr;   // stream of the last stage call (vz_engine_async_error resets that stream's stream-K tickets)
    int dec_nsplit = 1;          // context splits of the decode attention for the steps being launched
    float* d_logits = nullptr;   // [max_batch, vocab] fp32
    bf16_t* d_xnorm = nullptr;   // [64, hidden]: normalised rows of a 5..16-row decode batch (the MFMA weight stream reads them from L2)
    float* d_part = nullptr;     // decode attention partials
    unsigned* d_ticket = nullptr; // arrival counters of the fused decode attention
    unsigned* d_ao_done = nullptr; // arrival word of the attention + O-projection launch (attn_o_fused.hip); zeroed with the step counter
    int* d_ferr = nullptr;        // raised by a bounded device-side wait that expired
    int nsplit = 32;                 // upper bound: a split takes >= 128 keys, the splits beyond ceil(len / 128) leave at once
    VzTokState* tok = nullptr;          // persistent decode-token kernel (decode_persist.hip): per-layer pointer table + hand-off vectors + arrival counters
    bool use_tok = false;               // the steps being launched run on it (decided per vz_llm_decode_steps call)
    int tok_poke_word = -1; unsigned tok_poke_value = 0;      // TEST HOOK (vz_test_persist_poke): applied once, behind the next counter reset
    hipStream_t cap_stream = nullptr;   // stream capture is not allowed on the legacy null stream torch hands us
    hipGraphExec_t dec_graph = nullptr; int dec_graph_B = 0, dec_graph_n = 0, dec_graph_nsplit = 0, dec_graph_tok = -1; long dec_graph_samp[6] = {0, 0, 0, 0, 0, 0}; int* dec_graph_out = nullptr; char* dec_graph_arena = nullptr;
    int* h_pinned = nullptr;     // pinned staging for small host->device uploads
    size_t h_pinned_ints = 0;
    // profiling
    int prof_on = 0, prof_class = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev; size_t prof_used = 0;
};

static const void* W_(vz_engine* e, const std::string& name, int dtype, long n, int* rc) {
    auto it = e->w.find(name);
    if (it == e->w.end()) { vz_set_error("weight '%s' was never registered", name.c_str()); *rc = VZ_ERR_STATE; return nullptr; }
    if (it->second.dtype != dtype || it->second.n != n) {
        vz_set_error("weight '%s': expected dtype %d n %ld, got dtype %d n %ld", name.c_str(), dtype, n, it->second.dtype,
                     it->second.n);
        *rc = VZ_ERR_STATE;
        return nullptr;
    }
    return it->second.p;
}
#define WB(name, n) ((const bf16_t*)W_(e, name, 0, (long)(n), &rc))
#define WF(name, n) ((const float*)W_(e, name, 1, (long)(n), &rc))
// e4m3 copy + row scales of a decode-side weight (nullptr unless the engine was created with weight_fp8)
#define W8(name, n) (e->c.weight_fp8 ? (const unsigned char*)W_(e, name, 2, (long)(n), &rc) : (const unsigned char*)nullptr)
#define WS(name, n) (e->c.weight_fp8 ? (const float*)W_(e, name, 1, (long)(n), &rc) : (const float*)nullptr)
#define RC(expr) do { int _r = (expr); if (_r) return _r; } while (0)

struct ProfScope {
    vz_engine* e; hipStream_t s; bool on; size_t idx; bool ext = false;
    ProfScope(vz_engine* e_, int klass, hipStream_t s_) : e(e_), s(s_), on(false), idx(0) {
        if (e->prof_on && klass == e->prof_class) {
            if (e->prof_used == e->prof_ev.size()) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
                e->prof_ev.push_back({a, b});
            }
            idx = e->prof_used++;
            on = true;
            if (klass == K_GEMM || klass == K_GEMV) {      // stamped by the launch itself (kernel-only duration)
                g_vz_prof_start = e->prof_ev[idx].first;
                g_vz_prof_stop = e->prof_ev[idx].second;
                ext = true;
            } else {
                (void)hipEventRecord(e->prof_ev[idx].first, s);
            }
        }
    }
    ~ProfScope() {
        if (!on) return;
        if (ext) {
            if (g_vz_prof_start) { g_vz_prof_start = nullptr; g_vz_prof_stop = nullptr; e->prof_used--; }   // launch never happened
        } else {
            (void)hipEventRecord(e->prof_ev[idx].second, s);
        }
    }
};

static int linear(vz_engine* e, int klass_hint, const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M,
                  int N, int K, const float* bias, const bf16_t* res, int ldr, int act, int out_fp32, hipStream_t s,
                  const float* norm_w = nullptr, float norm_eps = 0.f, const unsigned char* W8 = nullptr, const float* ws = nullptr) {
    LinearArgs a = mk_linear(A, lda, W, ldw, C, ldc, M, N, K, bias, res, ldr, act, out_fp32);
    a.norm_w = norm_w; a.norm_eps = norm_eps; a.err = e->d_ferr; e->last_stream = s;
    if (klass_hint == 1 && M >= 2 && ldw == K) {       // decode step: the fragment-tiled copy of this weight, if the caller registered one
        if (e->tiled_dirty) {
            e->tiled.clear(); e->tiled8.clear();
            for (const auto& kv : e->w) {
                if ((kv.second.dtype != 0 && kv.second.dtype != 2) || kv.first.empty() || kv.first.back() != 't') continue;
                auto base = e->w.find(kv.first.substr(0, kv.first.size() - 1));
                if (base == e->w.end() || base->second.dtype != kv.second.dtype || base->second.n != kv.second.n) continue;
                if (kv.second.dtype == 0) e->tiled[base->second.p] = (const bf16_t*)kv.second.p;
                else e->tiled8[base->second.p] = (const unsigned char*)kv.second.p;
            }
            e->tiled_dirty = false;
        }
        auto it = e->tiled.find((const void*)W);
        if (it != e->tiled.end()) a.Wt = it->second;
        if (W8 && ws && M >= 17 && M <= 64) {
            // 17..64-row step of an e4m3-weight engine: every projection streams the e4m3 fragment-tiled copy (gemm_wide.hip, round 3) -
            // half the bytes of the bf16 routes these row counts took before (the row-major e4m3 stream of gemm_skinny.hip was slower than
            // bf16 at 17..32 rows and unused beyond).  The RMSNorm runs as its own launch.
            auto it8 = e->tiled8.find((const void*)W8);
            if (it8 != e->tiled8.end()) {
                LinearArgs t = a;
                t.W8t = it8->second; t.wscale = ws; t.W8 = nullptr; t.Wt = nullptr; t.wide_ok = true;
                if (norm_w) { t.A = e->d_xnorm; t.lda = K; t.norm_w = nullptr; }
                if (vz_wide_ok(t)) {
                    if (norm_w) {
                        ProfScope ps(e, K_NORM, s);
                        int r = vz_launch_rmsnorm(A, lda, e->d_xnorm, K, norm_w, M, K, norm_eps, s);
                        if (r) return r;
                    }
                    ProfScope ps(e, K_GEMV, s);
                    return vz_launch_wide(t, s);
                }
            }
        }
    }
    a.wide_ok = klass_hint == 1;          // 1 = decode step: rows are independent sequences
    if (W8 && ws) {                       // e4m3 copy of the same weights: only the weight-stream kernels (M <= 32) take it
        a.W8 = W8; a.wscale = ws;
        LinearArgs t = a;
        if (M > 16) t.norm_w = nullptr;   // 17..32 rows: the norm runs as its own kernel below
        if (!vz_gemv_ok(t) && !(g_skinny_mode && vz_skinny_ok(t))) { a.W8 = nullptr; a.wscale = nullptr; }
    }
    if (klass_hint == 1 && M >= 17 && M <= 64 && a.Wt && (!(W8 && ws) || M >= std::max(g_decode_tile_rows, 33))) {      // (e4m3 engines: from where they leave the e4m3 stream anyway)
        // 17..64-row decode step on the tiled weight copy (gemm_wide.hip): weights straight to registers, the activations of a 512-k
        // chunk staged once per 128 weight rows - for the projections whose row blocks fill the chip without a K split (gate|up,
        // lm_head: measured 52.7 vs 62 us and 55 vs 91 us at 64 rows; the split shapes stay on the tile GEMM).  The RMSNorm runs
        // as its own launch.
        LinearArgs t = a;
        t.W8 = nullptr; t.wscale = nullptr;          // (a weight_fp8 engine's bf16 tensors - and their tiled copies - hold the same dequantised values)
        if (norm_w) { t.A = e->d_xnorm; t.lda = K; t.norm_w = nullptr; }
        if (vz_wide_engine_ok(t)) {
            if (norm_w) {
                ProfScope ps(e, K_NORM, s);
                int r = vz_launch_rmsnorm(A, lda, e->d_xnorm, K, norm_w, M, K, norm_eps, s);
                if (r) return r;
            }
            ProfScope ps(e, K_GEMV, s);
            return vz_launch_wide(t, s);
        }
    }
    if (klass_hint == 1 && M >= ((W8 && ws) ? std::max(g_decode_tile_rows, 33) : g_decode_tile_rows) && M <= 64 && g_skinny_mode && (K & 63) == 0) {     // (e4m3 stream: ahead up to 32 rows)
        // 17..64-row decode step as a TILE GEMM: the 128^2 MFMA kernel streams every weight once at the rate its workgroups can pull
        // (gate-up 48 us = 4.9 TB/s whatever the row count), where the MFMA weight stream of gemm_skinny.hip re-reads the activations
        // per 16-row group and falls to 2.7 TB/s at 64 rows (tools/bench_rows.py, profiles/r02_rows.txt).  Projections with few column
        // tiles (QKV 48, O / down 32) are cut along K until ~512 workgroups are in flight.  The RMSNorm runs as its own launch.
        if (norm_w) {
            ProfScope ps(e, K_NORM, s);
            int r = vz_launch_rmsnorm(A, lda, e->d_xnorm, K, norm_w, M, K, norm_eps, s);
            if (r) return r;
            a.A = e->d_xnorm; a.lda = K; a.norm_w = nullptr;
        }
        a.W8 = nullptr; a.wscale = nullptr;          // (a weight_fp8 engine's bf16 tensors hold the same dequantised values)
        a.splitk_hint = vz_decode_splitk(N, K, act);
        ProfScope ps(e, K_GEMV, s);
        return vz_launch_gemm128(a, s);
    }
    if (norm_w && M > 4 && M <= 64 && K == e->c.hidden && g_skinny_mode && !(vz_skinny_ok(a) && vz_skinny_fused_norm_ok(a))) {
        // 5..16 rows without the persistent fused-norm kernel (knob 9 = 2, or a K it does not take): normalise once into an
        // L2-resident scratch and let the one-group-per-workgroup MFMA weight stream take its B fragments from there
        { ProfScope ps(e, K_NORM, s); int r = vz_launch_rmsnorm(A, lda, e->d_xnorm, K, norm_w, M, K, norm_eps, s); if (r) return r; }
        a.A = e->d_xnorm; a.lda = K; a.norm_w = nullptr;
        if (vz_skinny_ok(a)) { ProfScope ps(e, K_GEMV, s); return vz_launch_skinny(a, s); }
        a.A = A; a.lda = lda; a.norm_w = norm_w;
    }
    if (norm_w) { ProfScope ps(e, K_GEMV, s); return vz_launch_linear(a, s); }     // skinny MFMA stream or GEMV: both fuse the norm
    const bool gemv = vz_gemv_ok(a) || (g_skinny_mode && vz_skinny_ok(a));
    ProfScope ps(e, gemv ? K_GEMV : K_GEMM, s);
    return vz_launch_linear(a, s);
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off, cap; bool ok;
    Carver(char* b, size_t c) : base(b), off(0), cap(c), ok(true) {}
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = (T*)(base + off);
        off += n * sizeof(T);
        if (off > cap) ok = false;
        return p;
    }
};

static int ensure_arena(vz_engine* e, size_t bytes) {
    if (bytes <= e->arena_bytes) return VZ_OK;
    if (e->arena) { VZ_CHECK_HIP(hipDeviceSynchronize()); VZ_CHECK_HIP(hipFree(e->arena)); e->arena = nullptr; e->arena_bytes = 0; }
    VZ_CHECK_HIP(hipMalloc((void**)&e->arena, bytes));
    e->arena_bytes = bytes;
    return VZ_OK;
}

static int upload_ints(vz_engine* e, const int* h, size_t n, int* d, hipStream_t s) {
    // small host arrays go through a pinned staging buffer; the copy is enqueued on the stream
    if (n > e->h_pinned_ints) {
        if (e->h_pinned) { VZ_CHECK_HIP(hipStreamSynchronize(s)); VZ_CHECK_HIP(hipHostFree(e->h_pinned)); }
        VZ_CHECK_HIP(hipHostMalloc((void**)&e->h_pinned, n * sizeof(int)));
        e->h_pinned_ints = n;
    } else {
        VZ_CHECK_HIP(hipStreamSynchronize(s));  // previous use of the staging buffer has drained
    }
    memcpy(e->h_pinned, h, n * sizeof(int));
    VZ_CHECK_HIP(hipMemcpyAsync(d, e->h_pinned, n * sizeof(int), hipMemcpyHostToDevice, s));
    return VZ_OK;
}

extern "C" int vz_engine_create(const vz_config* cfg, vz_engine** out) {
    VZ_CHECK_ARG(cfg && out, "engine_create: null argument");
    const vz_config& c = *cfg;
    VZ_CHECK_ARG(c.head_dim == 128 && c.hidden % 512 == 0 && c.inter % 512 == 0 && c.n_heads % c.n_kv_heads == 0 &&
                     c.n_heads * c.head_dim == c.hidden, "engine_create: unsupported Zephyr geometry");
    VZ_CHECK_ARG(c.clip_hidden % 512 == 0 && c.clip_hidden / c.clip_heads == 64 && c.clip_image % c.clip_patch == 0,
                 "engine_create: unsupported CLIP geometry");
    VZ_CHECK_ARG(c.hidden / c.qf_heads == 512 && c.qf_kv_dim == (c.fusion_groups + 1) * c.clip_hidden && c.qf_queries == 32,
                 "engine_create: unsupported Q-Former geometry");
    VZ_CHECK_ARG(c.clip_layers + 1 >= c.fusion_groups * c.fusion_layers_per_group + 1, "engine_create: CLIP too shallow for the fusion");
    VZ_CHECK_ARG(c.max_batch >= 1 && c.max_ctx >= 64 && c.max_tiles >= 1 && c.max_text >= 0, "engine_create: bad capacity");
    VZ_CHECK_ARG(c.tp_size >= 1 && c.tp_rank >= 0 && c.tp_rank < c.tp_size, "engine_create: bad tp_size/tp_rank %d/%d", c.tp_size, c.tp_rank);
    // MLP shard: 16-row gate|up interleave and K %% 64 of the down-proj GEMM (tp 8: 14336 / 8 = 1792 columns; a shard that is not a
    // multiple of 512 takes the tile-GEMM path for the one-row down-proj instead of the GEMV)
    VZ_CHECK_ARG(c.n_kv_heads % c.tp_size == 0 && c.n_heads % c.tp_size == 0 && (c.inter / c.tp_size) % 64 == 0 && c.inter % c.tp_size == 0,
                 "engine_create: tp_size %d must divide the KV heads (%d) and leave MLP shards that are multiples of 64", c.tp_size, c.n_kv_heads);
    VZ_CHECK_ARG((c.n_heads / c.tp_size) == 4 * (c.n_kv_heads / c.tp_size), "engine_create: 4 query heads per KV head expected");
    { int r = vz_init_gemm_kernels(); if (r) return r; r = vz_init_attention_kernels(); if (r) return r; }
    vz_engine* e = new vz_engine();
    e->c = c;
    e->tp = c.tp_size; e->rank = c.tp_rank;
    e->Hq_l = c.n_heads / c.tp_size; e->Hkv_l = c.n_kv_heads / c.tp_size; e->I_l = c.inter / c.tp_size;
    e->Vp = (c.vocab + c.tp_size - 1) / c.tp_size;
    e->kv_layer_elems = (size_t)2 * c.max_batch * e->Hkv_l * c.max_ctx * c.head_dim;
    hipError_t er = hipMalloc((void**)&e->kv, e->kv_layer_elems * c.n_layers * sizeof(bf16_t));
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_state, (4 * c.max_batch + 4) * sizeof(int));
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_logits, (size_t)c.max_batch * e->Vp * e->tp * sizeof(float));
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_part, (size_t)c.max_batch * e->Hkv_l * 64 * (4 * 128 + 32) * sizeof(float));
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_ticket, 4096);
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_xnorm, (size_t)64 * c.hidden * sizeof(bf16_t));
    if (er == hipSuccess) er = hipMemset(e->d_ticket, 0, 4096);
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_ao_done, 64);
    if (er == hipSuccess) er = hipMemset(e->d_ao_done, 0, 64);
    if (er == hipSuccess) er = hipMalloc((void**)&e->d_ferr, sizeof(int));
    if (er == hipSuccess) er = hipMemset(e->d_ferr, 0, sizeof(int));
    if (er != hipSuccess) {
        vz_set_error("engine_create: hipMalloc failed: %s", hipGetErrorString(er));
        delete e;
        return VZ_ERR_HIP;
    }
    *out = e;
    return VZ_OK;
}

extern "C" int vz_engine_destroy(vz_engine* e) {
    if (!e) return VZ_OK;
    hipDeviceSynchronize();
    if (e->dec_graph) (void)hipGraphExecDestroy(e->dec_graph);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    for (auto& p : e->prof_ev) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    if (e->arena) hipFree(e->arena);
    if (e->kv) hipFree(e->kv);
    if (e->d_state) hipFree(e->d_state);
    if (e->d_logits) hipFree(e->d_logits);
    if (e->d_part) hipFree(e->d_part);
    if (e->d_ticket) hipFree(e->d_ticket);
    if (e->d_ao_done) hipFree(e->d_ao_done);
    if (e->d_ferr) hipFree(e->d_ferr);
    if (e->d_xnorm) hipFree(e->d_xnorm);
    if (e->d_gather) (void)hipFree(e->d_gather);
    vz_decode_persist_destroy(e->tok);
    if (e->comm) (void)ncclCommDestroy(e->comm);
    if (e->os_area) (void)hipFree(e->os_area);
    if (e->os_seq) (void)hipFree(e->os_seq);
    if (e->h_pinned) hipHostFree(e->h_pinned);
    delete e;
    return VZ_OK;
}

// The captured decode graph freezes every pointer its kernels take (weights, rotary tables, the gathered-logits buffer, the
// workspace): whoever replaces one of them drops the graph, the next vz_llm_decode_steps captures again.
static int drop_decode_graph(vz_engine* e) {
    if (e->tok) {                               // the persistent kernel's layer table freezes weight / cache pointers too
        VZ_CHECK_HIP(hipDeviceSynchronize());
        vz_decode_persist_destroy(e->tok);
        e->tok = nullptr;
    }
    if (!e->dec_graph) return VZ_OK;
    VZ_CHECK_HIP(hipDeviceSynchronize());       // a replay may still be running
    (void)hipGraphExecDestroy(e->dec_graph);
    e->dec_graph = nullptr;
    return VZ_OK;
}

extern "C" int vz_engine_set_weight(vz_engine* e, const char* name, const void* d_ptr, int dtype, long n_elems) {
    VZ_CHECK_ARG(e && name && d_ptr && dtype >= 0 && dtype <= 2 && n_elems > 0, "set_weight: bad argument");
    VZ_CHECK_ARG(((uintptr_t)d_ptr & 15) == 0, "set_weight: '%s' must be 16-byte aligned", name);
    RC(drop_decode_graph(e));
    e->w[name] = Weight{d_ptr, dtype, n_elems};
    e->finalized = false; e->tiled_dirty = true;
    return VZ_OK;
}

// forget a registered weight (the e4m3 copy / scales of a bf16 tensor that has been rewritten: finalize must not accept the stale ones)
extern "C" int vz_engine_unset_weight(vz_engine* e, const char* name) {
    VZ_CHECK_ARG(e && name, "unset_weight: bad argument");
    RC(drop_decode_graph(e));
    e->w.erase(name);
    e->finalized = false; e->tiled_dirty = true;
    return VZ_OK;
}

extern "C" int vz_engine_resize_vocab(vz_engine* e, int new_vocab) {
    VZ_CHECK_ARG(e && new_vocab > 0, "resize_vocab: bad argument");
    VZ_CHECK_ARG(e->tp == 1, "resize_vocab: not available on a tensor-parallel engine (rebuild it at the new vocabulary)");
    if (new_vocab == e->c.vocab) return VZ_OK;
    VZ_CHECK_HIP(hipDeviceSynchronize());
    if (e->dec_graph) { (void)hipGraphExecDestroy(e->dec_graph); e->dec_graph = nullptr; }     // it holds the old logits width
    if (e->tok) { vz_decode_persist_destroy(e->tok); e->tok = nullptr; }
    if (e->d_logits) { VZ_CHECK_HIP(hipFree(e->d_logits)); e->d_logits = nullptr; }
    e->c.vocab = new_vocab;
    e->Vp = new_vocab;
    VZ_CHECK_HIP(hipMalloc((void**)&e->d_logits, (size_t)e->c.max_batch * e->Vp * sizeof(float)));
    e->w.erase("llm.embed");
    e->w.erase("llm.lm_head");
    e->finalized = false;
    return VZ_OK;
}

static int kpad_patch(const vz_config& c) { return (int)align_up((size_t)3 * c.clip_patch * c.clip_patch, 64); }

extern "C" int vz_engine_finalize(vz_engine* e) {
    VZ_CHECK_ARG(e, "finalize: null engine");
    const vz_config& c = e->c;
    int rc = VZ_OK;
    const long C = c.clip_hidden, H = c.hidden, tokens = (c.clip_image / c.clip_patch) * (c.clip_image / c.clip_patch) + 1;
    WB("clip.patch_w", C * kpad_patch(c)); WB("clip.cls", C); WB("clip.pos", tokens * C);
    WF("clip.pre_ln.w", C); WF("clip.pre_ln.b", C);
    for (int i = 0; i < c.clip_layers && !rc; ++i) {
        const std::string p = "clip." + std::to_string(i) + ".";
        WF(p + "ln1.w", C); WF(p + "ln1.b", C); WB(p + "qkv.w", 3 * C * C); WF(p + "qkv.b", 3 * C);
        WB(p + "o.w", C * C); WF(p + "o.b", C); WF(p + "ln2.w", C); WF(p + "ln2.b", C);
        WB(p + "fc1.w", (long)c.clip_inter * C); WF(p + "fc1.b", c.clip_inter);
        WB(p + "fc2.w", (long)c.clip_inter * C); WF(p + "fc2.b", C);
    }
    WB("qf.queries", (long)c.qf_queries * H); WF("qf.pre_norm.w", c.qf_kv_dim); WF("qf.pre_norm.b", c.qf_kv_dim);
    WF("qf.norm.w", H); WF("qf.norm.b", H);
    for (int i = 0; i < c.qf_blocks && !rc; ++i) {
        const std::string p = "qf." + std::to_string(i) + ".";
        for (const char* n : {"n1", "n2", "n3"}) { WF(p + n + ".w", H); WF(p + n + ".b", H); }
        WB(p + "sa_in.w", 3 * H * H); WF(p + "sa_in.b", 3 * H); WB(p + "sa_out.w", H * H); WF(p + "sa_out.b", H);
        WB(p + "ca_q.w", H * H); WF(p + "ca_q.b", H); WB(p + "ca_kv.w", 2 * H * c.qf_kv_dim); WF(p + "ca_kv.b", 2 * H);
        WB(p + "ca_out.w", H * H); WF(p + "ca_out.b", H);
        WB(p + "ffn1.w", 2 * H * H); WF(p + "ffn1.b", 2 * H); WB(p + "ffn2.w", 2 * H * H); WF(p + "ffn2.b", H);
    }
    const long qkv_n = (long)(e->Hq_l + 2 * e->Hkv_l) * c.head_dim;
    WB("llm.embed", (long)c.vocab * H); WF("llm.norm", H); WB("llm.lm_head", (long)e->Vp * H);
    W8("llm.lm_head8", (long)e->Vp * H); WS("llm.lm_heads", e->Vp);
    for (int i = 0; i < c.n_layers && !rc; ++i) {
        const std::string p = "llm." + std::to_string(i) + ".";
        WF(p + "in_norm", H); WF(p + "post_norm", H); WB(p + "qkv.w", qkv_n * H); WB(p + "o.w", H * (long)e->Hq_l * c.head_dim);
        WB(p + "gu.w", 2L * e->I_l * H); WB(p + "down.w", (long)e->I_l * H);
        W8(p + "qkv.w8", qkv_n * H); WS(p + "qkv.ws", qkv_n); W8(p + "o.w8", H * (long)e->Hq_l * c.head_dim); WS(p + "o.ws", H);
        W8(p + "gu.w8", 2L * e->I_l * H); WS(p + "gu.ws", 2L * e->I_l); W8(p + "down.w8", (long)e->I_l * H); WS(p + "down.ws", H);
    }
    if (rc) return rc;
    e->finalized = true;
    return VZ_OK;
}

extern "C" int vz_engine_set_rope(vz_engine* e, const float* d_cos, const float* d_sin, int max_pos) {
    VZ_CHECK_ARG(e && d_cos && d_sin && max_pos > 0, "set_rope: bad argument");
    RC(drop_decode_graph(e));
    e->cosT = d_cos; e->sinT = d_sin; e->rope_max = max_pos;
    return VZ_OK;
}

#define NEED_READY()                                                                                         \
    do {                                                                                                     \
        if (!e || !e->finalized) { vz_set_error("engine not finalized (register every weight, then vz_engine_finalize)"); return VZ_ERR_STATE; } \
    } while (0)

// ------------------------------------------------------------------------------------------------
// a8-a10: CLIP tower + fusion
// ------------------------------------------------------------------------------------------------
extern "C" int vz_clip_fused_features(vz_engine* e, const void* d_images, int T, void* d_out, void* d_hidden_dbg,
                                      vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(d_images && d_out && T >= 1 && T <= c.max_tiles, "clip: T=%d outside [1,%d]", T, c.max_tiles);
    const int C = c.clip_hidden, g = c.clip_image / c.clip_patch, P = g * g, tokens = P + 1, kpad = kpad_patch(c);
    const int L = c.clip_layers, rows = T * tokens;
    const size_t hs_layer = (size_t)rows * C;
    size_t need = 0;
    {
        Carver m(nullptr, ~(size_t)0);
        m.take<bf16_t>((size_t)T * P * kpad); m.take<bf16_t>((size_t)T * P * C);
        if (!d_hidden_dbg) m.take<bf16_t>(hs_layer * (L + 1));
        m.take<bf16_t>(hs_layer); m.take<bf16_t>(hs_layer * 3); m.take<bf16_t>(hs_layer); m.take<bf16_t>((size_t)rows * c.clip_inter);
        need = m.off + 256;
    }
    RC(ensure_arena(e, need));
    Carver m(e->arena, e->arena_bytes);
    bf16_t* col = m.take<bf16_t>((size_t)T * P * kpad);
    bf16_t* pe = m.take<bf16_t>((size_t)T * P * C);
    bf16_t* hs = d_hidden_dbg ? (bf16_t*)d_hidden_dbg : m.take<bf16_t>(hs_layer * (L + 1));
    bf16_t* y = m.take<bf16_t>(hs_layer);
    bf16_t* qkv = m.take<bf16_t>(hs_layer * 3);
    bf16_t* att = m.take<bf16_t>(hs_layer);
    bf16_t* mlp = m.take<bf16_t>((size_t)rows * c.clip_inter);
    int rc = VZ_OK;
    {
        ProfScope ps(e, K_OTHER, s);
        RC(vz_launch_im2col((const bf16_t*)d_images, T, c.clip_image, c.clip_patch, kpad, col, s));
    }
    RC(linear(e, 0, col, kpad, WB("clip.patch_w", (long)C * kpad), kpad, pe, C, T * P, C, kpad, nullptr, nullptr, 0, VZ_ACT_NONE, 0, s));
    {
        ProfScope ps(e, K_OTHER, s);
        RC(vz_launch_clip_assemble(pe, WB("clip.cls", C), WB("clip.pos", (long)tokens * C), T, tokens, C, att, s));
    }
    {
        ProfScope ps(e, K_NORM, s);
        RC(vz_launch_layernorm(att, C, hs, C, WF("clip.pre_ln.w", C), WF("clip.pre_ln.b", C), rows, C, c.clip_eps, s));
    }
    for (int i = 0; i < L; ++i) {
        const std::string p = "clip." + std::to_string(i) + ".";
        bf16_t* x = hs + hs_layer * i;
        bf16_t* xn = hs + hs_layer * (i + 1);
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x, C, y, C, WF(p + "ln1.w", C), WF(p + "ln1.b", C), rows, C, c.clip_eps, s)); }
        RC(linear(e, 0, y, C, WB(p + "qkv.w", 3L * C * C), C, qkv, 3 * C, rows, 3 * C, C, WF(p + "qkv.b", 3 * C), nullptr, 0, VZ_ACT_NONE, 0, s));
        {
            ProfScope ps(e, K_ATTN, s);
            AttnArgs a;
            a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C; a.o = att;
            a.B = T; a.Sq = tokens; a.Sk = tokens; a.Hq = c.clip_heads; a.Hkv = c.clip_heads; a.head_dim = 64;
            a.q_bs = a.k_bs = a.v_bs = (long)tokens * 3 * C; a.q_ss = a.k_ss = a.v_ss = 3 * C; a.q_hs = a.k_hs = a.v_hs = 64;
            a.o_bs = (long)tokens * C; a.o_ss = C; a.o_hs = 64;
            a.scale = 0.125f; a.causal = 0; a.q_pos0 = 0; a.window = 0; a.kv_len = nullptr;
            RC(vz_launch_attention(a, s));
        }
        RC(linear(e, 0, att, C, WB(p + "o.w", (long)C * C), C, xn, C, rows, C, C, WF(p + "o.b", C), x, C, VZ_ACT_NONE, 0, s));
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(xn, C, y, C, WF(p + "ln2.w", C), WF(p + "ln2.b", C), rows, C, c.clip_eps, s)); }
        RC(linear(e, 0, y, C, WB(p + "fc1.w", (long)c.clip_inter * C), C, mlp, c.clip_inter, rows, c.clip_inter, C,
                  WF(p + "fc1.b", c.clip_inter), nullptr, 0, VZ_ACT_QUICK_GELU, 0, s));
        RC(linear(e, 0, mlp, c.clip_inter, WB(p + "fc2.w", (long)c.clip_inter * C), c.clip_inter, xn, C, rows, C, c.clip_inter,
                  WF(p + "fc2.b", C), xn, C, VZ_ACT_NONE, 0, s));
        if (rc) return rc;
    }
    {
        ProfScope ps(e, K_OTHER, s);
        const int first = L - c.fusion_groups * c.fusion_layers_per_group;
        RC(vz_launch_fusion(hs, (long)hs_layer, first, c.fusion_groups, c.fusion_layers_per_group, T, tokens, C, c.clip_keep_cls ? 0 : 1, (bf16_t*)d_out, s));
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// a11: Q-Former
// ------------------------------------------------------------------------------------------------
static int g_qf_kv_all = 1;     // vz_tune_set(25, 0): the Q-Former's cross-attention K|V projections one block at a time (A/B)
static int qf_attn(vz_engine* e, const bf16_t* q, long q_bs, long q_ss, const bf16_t* k, const bf16_t* v, long kv_bs, long kv_ss,
                   bf16_t* o, int B, int Sq, int Sk, hipStream_t s, float* part = nullptr, size_t part_floats = 0) {
    ProfScope ps(e, K_ATTN, s);
    AttnArgs a;
    a.part = part; a.part_floats = part_floats;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.B = B; a.Sq = Sq; a.Sk = Sk; a.Hq = e->c.qf_heads; a.Hkv = e->c.qf_heads; a.head_dim = 512;
    a.q_bs = q_bs; a.q_ss = q_ss; a.q_hs = 512; a.k_bs = a.v_bs = kv_bs; a.k_ss = a.v_ss = kv_ss; a.k_hs = a.v_hs = 512;
    a.o_bs = (long)Sq * e->c.hidden; a.o_ss = e->c.hidden; a.o_hs = 512;
    a.scale = 0.044194173824159216f;  // 512^-0.5
    a.causal = 0; a.q_pos0 = 0; a.window = 0; a.kv_len = nullptr;
    return vz_launch_attention(a, s);
}

extern "C" int vz_qformer(vz_engine* e, const void* d_feats, int T, const void* d_text, int n_samples, int Lmax,
                          const int* h_tile_sample, void* d_out, vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(d_feats && d_out && T >= 1 && T <= c.max_tiles, "qformer: T=%d outside [1,%d]", T, c.max_tiles);
    VZ_CHECK_ARG(Lmax >= 0 && Lmax <= c.max_text, "qformer: Lmax=%d outside [0,%d]", Lmax, c.max_text);
    VZ_CHECK_ARG(n_samples >= 1 && n_samples <= T && h_tile_sample && (Lmax == 0 || d_text), "qformer: bad sample map");
    for (int t = 0; t < T; ++t) VZ_CHECK_ARG(h_tile_sample[t] >= 0 && h_tile_sample[t] < n_samples, "qformer: tile_sample[%d] out of range", t);
    const int H = c.hidden, NQ = c.qf_queries, KD = c.qf_kv_dim;
    const int P = (c.clip_image / c.clip_patch) * (c.clip_image / c.clip_patch) + (c.clip_keep_cls ? 1 : 0);   // visual tokens per tile
    const int N0 = NQ + Lmax, FF = 2 * H;
    const size_t R = (size_t)T * NQ;  // query rows in flight after block 0's self-attention
    int rc = VZ_OK;
    // The cross-attention K|V projections of all blocks read the same pre-normed features.  When their weights (and biases) lie back
    // to back in memory (vz_hip/engine.py allocates them so; a C-ABI caller may too) they run as ONE product over N = blocks * 2H:
    // 256 column tiles per row tile = whole residency rounds of the 256^2 kernel, no stream-K tail (8 x 217 us -> one launch).  The
    // [T * P, blocks * 2H] result is capped at 2 GiB; larger tile batches fill the chip per block anyway.
    const int nb = c.qf_blocks;
    const bf16_t* kvw0 = WB("qf.0.ca_kv.w", 2L * H * KD);
    const float* kvb0 = WF("qf.0.ca_kv.b", 2 * H);
    bool kv_all = nb > 1 && (size_t)T * P * nb * 2 * H * sizeof(bf16_t) <= ((size_t)2 << 30) && g_qf_kv_all;
    for (int i = 1; i < nb && kv_all && !rc; ++i) {
        const std::string p = "qf." + std::to_string(i) + ".";
        kv_all = WB(p + "ca_kv.w", 2L * H * KD) == kvw0 + (size_t)i * 2 * H * KD && WF(p + "ca_kv.b", 2 * H) == kvb0 + (size_t)i * 2 * H;
    }
    if (rc) return rc;
    const int kv_ld = kv_all ? nb * 2 * H : 2 * H;
    size_t need;
    {
        Carver m(nullptr, ~(size_t)0);
        m.take<bf16_t>((size_t)T * P * KD); m.take<bf16_t>((size_t)T * P * kv_ld);
        m.take<bf16_t>((size_t)n_samples * N0 * H); m.take<bf16_t>((size_t)n_samples * N0 * H); m.take<bf16_t>((size_t)n_samples * N0 * 2 * H);
        m.take<bf16_t>((size_t)NQ * H); m.take<bf16_t>((size_t)n_samples * NQ * H); m.take<bf16_t>((size_t)n_samples * NQ * H);
        m.take<bf16_t>(R * H); m.take<bf16_t>(R * H); m.take<bf16_t>(R * 3 * H); m.take<bf16_t>(R * H); m.take<bf16_t>(R * FF);
        m.take<float>(std::max((size_t)T * ((P + 95) / 96), (size_t)n_samples * ((N0 + 95) / 96)) * c.qf_heads * NQ * (512 + 4));
        need = m.off + 256;
    }
    RC(ensure_arena(e, need));
    Carver m(e->arena, e->arena_bytes);
    bf16_t* fn = m.take<bf16_t>((size_t)T * P * KD);            // pre_norm(features)
    bf16_t* ckv = m.take<bf16_t>((size_t)T * P * kv_ld);        // cross-attention K|V of the current block (kv_all: of every block)
    bf16_t* x0 = m.take<bf16_t>((size_t)n_samples * N0 * H);    // [queries ; text] per sample
    bf16_t* y0 = m.take<bf16_t>((size_t)n_samples * N0 * H);
    bf16_t* kv0 = m.take<bf16_t>((size_t)n_samples * N0 * 2 * H);
    bf16_t* q0 = m.take<bf16_t>((size_t)NQ * H);
    bf16_t* a0 = m.take<bf16_t>((size_t)n_samples * NQ * H);
    bf16_t* xs = m.take<bf16_t>((size_t)n_samples * NQ * H);
    bf16_t* x = m.take<bf16_t>(R * H);
    bf16_t* y = m.take<bf16_t>(R * H);
    bf16_t* qkv = m.take<bf16_t>(R * 3 * H);
    bf16_t* att = m.take<bf16_t>(R * H);
    bf16_t* ff = m.take<bf16_t>(R * FF);
    // key-split partials (96 keys per workgroup) of the cross-attention and of block 0's self-attention over [queries ; text]
    const size_t part_floats = std::max((size_t)T * ((P + 95) / 96), (size_t)n_samples * ((N0 + 95) / 96)) * c.qf_heads * NQ * (512 + 4);
    float* part = m.take<float>(part_floats);
    { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm((const bf16_t*)d_feats, KD, fn, KD, WF("qf.pre_norm.w", KD), WF("qf.pre_norm.b", KD), T * P, KD, c.qf_eps, s)); }
    const bf16_t* queries = WB("qf.queries", (long)NQ * H);
    if (rc) return rc;
    // ---- block 0 self-attention, once per sample, query rows only ----
    {
        ProfScope ps(e, K_OTHER, s);
        for (int sm = 0; sm < n_samples; ++sm) {
            RC(vz_launch_copy_rows(queries, H, x0 + (size_t)sm * N0 * H, H, NQ, H, s));
            if (Lmax > 0) RC(vz_launch_copy_rows((const bf16_t*)d_text + (size_t)sm * Lmax * H, H, x0 + ((size_t)sm * N0 + NQ) * H, H, Lmax, H, s));
            RC(vz_launch_copy_rows(queries, H, xs + (size_t)sm * NQ * H, H, NQ, H, s));
        }
    }
    {
        const std::string p = "qf.0.";
        const bf16_t* w_in = WB(p + "sa_in.w", 3L * H * H);
        const float* b_in = WF(p + "sa_in.b", 3 * H);
        if (rc) return rc;
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x0, H, y0, H, WF(p + "n1.w", H), WF(p + "n1.b", H), n_samples * N0, H, c.qf_eps, s)); }
        RC(linear(e, 0, y0, H, w_in + (size_t)H * H, H, kv0, 2 * H, n_samples * N0, 2 * H, H, b_in + H, nullptr, 0, VZ_ACT_NONE, 0, s));
        RC(linear(e, 0, y0, H, w_in, H, q0, H, NQ, H, H, b_in, nullptr, 0, VZ_ACT_NONE, 0, s));  // rows 0..31 of sample 0 = LN1(queries)
        RC(qf_attn(e, q0, 0, H, kv0, kv0 + H, (long)N0 * 2 * H, 2 * H, a0, n_samples, NQ, N0, s, part, part_floats));
        RC(linear(e, 0, a0, H, WB(p + "sa_out.w", (long)H * H), H, xs, H, n_samples * NQ, H, H, WF(p + "sa_out.b", H), xs, H, VZ_ACT_NONE, 0, s));
        ProfScope ps(e, K_OTHER, s);
        for (int t = 0; t < T; ++t) RC(vz_launch_copy_rows(xs + (size_t)h_tile_sample[t] * NQ * H, H, x + (size_t)t * NQ * H, H, NQ, H, s));
    }
    for (int i = 0; i < c.qf_blocks; ++i) {
        const std::string p = "qf." + std::to_string(i) + ".";
        if (i > 0) {
            { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x, H, y, H, WF(p + "n1.w", H), WF(p + "n1.b", H), (int)R, H, c.qf_eps, s)); }
            RC(linear(e, 0, y, H, WB(p + "sa_in.w", 3L * H * H), H, qkv, 3 * H, (int)R, 3 * H, H, WF(p + "sa_in.b", 3 * H), nullptr, 0, VZ_ACT_NONE, 0, s));
            RC(qf_attn(e, qkv, (long)NQ * 3 * H, 3 * H, qkv + H, qkv + 2 * H, (long)NQ * 3 * H, 3 * H, att, T, NQ, NQ, s));
            RC(linear(e, 0, att, H, WB(p + "sa_out.w", (long)H * H), H, x, H, (int)R, H, H, WF(p + "sa_out.b", H), x, H, VZ_ACT_NONE, 0, s));
        }
        // cross-attention against the tile's 576 fused visual tokens
        if (!kv_all) RC(linear(e, 0, fn, KD, WB(p + "ca_kv.w", 2L * H * KD), KD, ckv, 2 * H, T * P, 2 * H, KD, WF(p + "ca_kv.b", 2 * H), nullptr, 0, VZ_ACT_NONE, 0, s));
        else if (i == 0) RC(linear(e, 0, fn, KD, kvw0, KD, ckv, kv_ld, T * P, kv_ld, KD, kvb0, nullptr, 0, VZ_ACT_NONE, 0, s));
        const bf16_t* ck = kv_all ? ckv + (size_t)i * 2 * H : ckv;
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x, H, y, H, WF(p + "n2.w", H), WF(p + "n2.b", H), (int)R, H, c.qf_eps, s)); }
        RC(linear(e, 0, y, H, WB(p + "ca_q.w", (long)H * H), H, qkv, H, (int)R, H, H, WF(p + "ca_q.b", H), nullptr, 0, VZ_ACT_NONE, 0, s));
        RC(qf_attn(e, qkv, (long)NQ * H, H, ck, ck + H, (long)P * kv_ld, kv_ld, att, T, NQ, P, s, part, part_floats));
        RC(linear(e, 0, att, H, WB(p + "ca_out.w", (long)H * H), H, x, H, (int)R, H, H, WF(p + "ca_out.b", H), x, H, VZ_ACT_NONE, 0, s));
        // FFN
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x, H, y, H, WF(p + "n3.w", H), WF(p + "n3.b", H), (int)R, H, c.qf_eps, s)); }
        RC(linear(e, 0, y, H, WB(p + "ffn1.w", 2L * H * H), H, ff, FF, (int)R, FF, H, WF(p + "ffn1.b", FF), nullptr, 0, VZ_ACT_GELU_ERF, 0, s));
        RC(linear(e, 0, ff, FF, WB(p + "ffn2.w", 2L * H * H), FF, x, H, (int)R, H, FF, WF(p + "ffn2.b", H), x, H, VZ_ACT_NONE, 0, s));
        if (rc) return rc;
    }
    { ProfScope ps(e, K_NORM, s); RC(vz_launch_layernorm(x, H, (bf16_t*)d_out, H, WF("qf.norm.w", H), WF("qf.norm.b", H), (int)R, H, c.qf_eps, s)); }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// a6/a7: splice
// ------------------------------------------------------------------------------------------------
extern "C" int vz_embed_splice(vz_engine* e, const int* d_kind, const int* d_idx, int rows, const void* d_visual, void* d_out,
                               vz_stream stream) {
    NEED_READY();
    int rc = VZ_OK;
    const bf16_t* table = WB("llm.embed", (long)e->c.vocab * e->c.hidden);
    if (rc) return rc;
    VZ_CHECK_ARG(d_idx && d_out && rows > 0, "splice: bad argument");
    ProfScope ps(e, K_OTHER, (hipStream_t)stream);
    return vz_launch_gather_rows(d_kind, d_idx, rows, e->c.hidden, table, (const bf16_t*)d_visual, (bf16_t*)d_out, (hipStream_t)stream);
}


// ---- tensor-parallel collectives (RCCL over xGMI); no-ops at tp == 1 ----
// Self-test (vz_tune_set(7, 1)): an engine with tp_size == 1 that has been given a one-rank communicator routes the same
// call sites through RCCL (all-reduce over one rank = identity, all-gather = copy), so the collective plumbing - library,
// dtypes, in-place buffers, stream order, the vocab-parallel gather + repack - runs on a single GPU.
static int g_force_comm = 0;
static int g_rope_in_attn = 1;     // vz_tune_set(33, 0): the prefill writes a rotated copy of Q (rope_kv_kernel) for the attention again (A/B; bit-identical)
static int g_flash_bwd = 1;               // vz_tune_set(32, 0): the training step's head-128 attention backward through the materialising batched-GEMM route again (A/B; train_engine.inc)
static int g_attn_o = 1;           // vz_tune_set(30, 0): batch-1 decode attention and O projection as two launches again (attn_o_fused.hip off)
static int g_persist_decode = 0;   // vz_tune_set(28, 1): batch-1 decode steps as one resident grid per token (decode_persist.hip) instead of the launch chain.
                                   // Off by default: measured 282 vs 339 tok/s (profiles/r03_persist_stamps.txt: the phase edges + the attention phase leave HBM idle longer than the launch boundaries they replace)
static int g_attn_nsplit = 0;   // vz_tune_set(10, n): context splits of the fused decode attention (0 = engine default)
static inline bool tp_local(const vz_engine* e) { return e->tp == 1 && !(g_force_comm && e->comm); }

// vz_tune_set(7, 2): shape rehearsal of ONE rank of a tp_size > 1 engine on a single GPU - every collective is skipped (the
// partial sums / vocab shard are left as they are), so all local kernels run with that rank's shard shapes; results are
// meaningless as logits, the point is that nothing on the local path rejects the shapes of tp 2 / 4 / 8.
static inline bool tp_skip(const vz_engine* e) { return e->tp > 1 && g_force_comm == 2; }

static int g_oneshot = 1;        // vz_tune_set(29, 0): keep RCCL for the decode step's all-reduces although peer areas are attached
static int tp_allreduce_bf16(vz_engine* e, bf16_t* buf, size_t count, hipStream_t s, bool decode = false) {
    if (decode && g_oneshot && e->os_ranks == e->tp && e->tp > 1 && (int)count <= vz_engine::OS_MAX_ELEMS && (count & 1) == 0 && !tp_skip(e)) {
        ProfScope ps(e, K_COMM, s);
        return vz_launch_allreduce_oneshot(e->os_areas, e->rank, e->tp, vz_engine::OS_MAX_ELEMS, buf, buf, (int)count, e->os_seq, e->d_ferr, s);
    }
    if (tp_local(e) || tp_skip(e)) return VZ_OK;
    if (!e->comm) { vz_set_error("tensor-parallel engine used before vz_comm_init"); return VZ_ERR_STATE; }
    ProfScope ps(e, K_COMM, s);
    ncclResult_t r = ncclAllReduce(buf, buf, count, ncclBfloat16, ncclSum, e->comm, s);
    if (r != ncclSuccess) { vz_set_error("ncclAllReduce failed: %s", ncclGetErrorString(r)); return VZ_ERR_HIP; }
    return VZ_OK;
}

// lm_head over `rows` hidden rows -> fp32 logits [rows, vocab] in `out`.  Vocab-parallel under TP: every rank computes its
// Vp rows of the table, the shards are all-gathered and repacked to the dense [rows, vocab] layout on every rank.
// vocab-parallel lm_head: [tp][rows][Vp] gathered logits + this rank's shard behind them (never grows inside a stream capture:
// vz_llm_decode_steps sizes it before it captures)
static int ensure_gather(vz_engine* e, int rows, hipStream_t s) {
    if (e->tp == 1 && !e->comm) return VZ_OK;
    const size_t local_off = ((size_t)e->tp * rows * e->Vp + 3) & ~(size_t)3;
    const size_t need = local_off + (size_t)rows * e->Vp;
    if (need > e->gather_floats) {
        RC(drop_decode_graph(e));    // its all-gather / repack / lm_head nodes hold the old buffer and the old shard offset
        if (e->d_gather) { VZ_CHECK_HIP(hipStreamSynchronize(s)); VZ_CHECK_HIP(hipFree(e->d_gather)); e->d_gather = nullptr; }
        VZ_CHECK_HIP(hipMalloc((void**)&e->d_gather, need * sizeof(float)));
        e->gather_floats = need;
    }
    return VZ_OK;
}

static int lm_head_logits(vz_engine* e, const bf16_t* h, int rows, float* out, hipStream_t s, const float* norm_w) {
    const vz_config& c = e->c;
    const int H = c.hidden;
    int rc = VZ_OK;
    const bf16_t* lm = WB("llm.lm_head", (long)e->Vp * H);
    if (rc) return rc;
    const unsigned char* lm8 = W8("llm.lm_head8", (long)e->Vp * H);
    const float* lms = WS("llm.lm_heads", e->Vp);
    if (rc) return rc;
    if (tp_local(e)) return linear(e, norm_w ? 1 : 0, h, H, lm, H, out, c.vocab, rows, c.vocab, H, nullptr, nullptr, 0, VZ_ACT_NONE, 1, s, norm_w, c.rms_eps, lm8, lms);
    const size_t local_off = ((size_t)e->tp * rows * e->Vp + 3) & ~(size_t)3;      // the GEMM wants a 16-byte-aligned output base
    { int r = ensure_gather(e, rows, s); if (r) return r; }
    float* local = e->d_gather + local_off;
    RC(linear(e, norm_w ? 1 : 0, h, H, lm, H, local, e->Vp, rows, e->Vp, H, nullptr, nullptr, 0, VZ_ACT_NONE, 1, s, norm_w, c.rms_eps, lm8, lms));
    if (!e->comm && !tp_skip(e)) { vz_set_error("tensor-parallel engine used before vz_comm_init"); return VZ_ERR_STATE; }
    if (tp_skip(e)) {     // rehearsal: this rank's shard goes to its own chunk, the others stay zero
        VZ_CHECK_HIP(hipMemsetAsync(e->d_gather, 0, (size_t)e->tp * rows * e->Vp * sizeof(float), s));
        VZ_CHECK_HIP(hipMemcpyAsync(e->d_gather + (size_t)e->rank * rows * e->Vp, local, (size_t)rows * e->Vp * sizeof(float), hipMemcpyDeviceToDevice, s));
        return vz_launch_repack_logits(e->d_gather, out, rows, e->Vp, c.vocab, e->tp, s);
    }
    ncclResult_t r;
    { ProfScope ps(e, K_COMM, s); r = ncclAllGather(local, e->d_gather, (size_t)rows * e->Vp, ncclFloat, e->comm, s); }
    if (r != ncclSuccess) { vz_set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return VZ_ERR_HIP; }
    return vz_launch_repack_logits(e->d_gather, out, rows, e->Vp, c.vocab, e->tp, s);
}

extern "C" int vz_comm_unique_id(char* out128) {
    VZ_CHECK_ARG(out128, "comm_unique_id: null buffer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) { vz_set_error("ncclGetUniqueId failed: %s", ncclGetErrorString(r)); return VZ_ERR_HIP; }
    memcpy(out128, &id, 128);
    return VZ_OK;
}

extern "C" int vz_comm_init(vz_engine* e, const char* id128) {
    VZ_CHECK_ARG(e && id128, "comm_init: null argument");
    if (e->comm) return VZ_OK;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = ncclCommInitRank(&e->comm, e->tp, id, e->rank);
    if (r != ncclSuccess) { e->comm = nullptr; vz_set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r)); return VZ_ERR_HIP; }
    return VZ_OK;
}

extern "C" size_t vz_op_oneshot_area_bytes(int n_ranks, int max_elems) { return vz_oneshot_area_bytes(n_ranks, max_elems); }
extern "C" int vz_op_allreduce_oneshot(void* const* d_areas, int rank, int n_ranks, int max_elems, const void* d_in, void* d_out, int n, unsigned* d_seq,
                                       int* d_err, vz_stream stream) {
    return vz_launch_allreduce_oneshot(d_areas, rank, n_ranks, max_elems, (const bf16_t*)d_in, (bf16_t*)d_out, n, d_seq, d_err, (hipStream_t)stream);
}
// TEST FORM: the n_ranks ranks of one all-reduce as slices of ONE grid (one process, one GPU: co-resident whatever the stream / queue mapping)
extern "C" int vz_test_allreduce_oneshot_all(void* const* d_areas, int n_ranks, int max_elems, const void* const* d_in, void* const* d_out, int n,
                                             unsigned* const* d_seq, int* d_err, vz_stream stream) {
    return vz_launch_allreduce_oneshot_all(d_areas, n_ranks, max_elems, (const bf16_t* const*)d_in, (bf16_t* const*)d_out, n, d_seq, d_err, (hipStream_t)stream);
}
extern "C" int vz_comm_oneshot_local(vz_engine* e, void** out_area, size_t* out_bytes) {
    VZ_CHECK_ARG(e && out_area && out_bytes && e->tp >= 1, "comm_oneshot_local: bad argument");
    const size_t bytes = vz_oneshot_area_bytes(e->tp, vz_engine::OS_MAX_ELEMS);
    if (!e->os_area) {
        VZ_CHECK_HIP(hipMalloc(&e->os_area, bytes));
        VZ_CHECK_HIP(hipMemset(e->os_area, 0, bytes));
        VZ_CHECK_HIP(hipMalloc((void**)&e->os_seq, 2 * sizeof(unsigned)));
        const unsigned init[2] = {1u, 0u};                  // sequence numbers start at 1: a zero-filled area carries tag 0 = "nothing yet"
        VZ_CHECK_HIP(hipMemcpy(e->os_seq, init, sizeof(init), hipMemcpyHostToDevice));
    }
    *out_area = e->os_area; *out_bytes = bytes;
    return VZ_OK;
}
extern "C" int vz_comm_oneshot_attach(vz_engine* e, void* const* d_areas, int n_ranks) {
    VZ_CHECK_ARG(e && d_areas && n_ranks == e->tp && n_ranks <= 8 && e->os_area, "comm_oneshot_attach: needs tp_size areas after vz_comm_oneshot_local");
    VZ_CHECK_ARG(d_areas[e->rank] == e->os_area, "comm_oneshot_attach: area %d must be this rank's own", e->rank);
    RC(drop_decode_graph(e));
    for (int q = 0; q < n_ranks; ++q) { VZ_CHECK_ARG(d_areas[q], "comm_oneshot_attach: area %d missing", q); e->os_areas[q] = d_areas[q]; }
    e->os_ranks = n_ranks;
    return VZ_OK;
}

extern "C" int vz_tp_all_gather(vz_engine* e, const void* d_send, void* d_recv, size_t bytes_per_rank, vz_stream stream) {
    VZ_CHECK_ARG(e && d_send && d_recv && bytes_per_rank > 0, "tp_all_gather: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (tp_local(e)) { VZ_CHECK_HIP(hipMemcpyAsync(d_recv, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, s)); return VZ_OK; }
    if (tp_skip(e)) {     // rehearsal: own chunk only
        VZ_CHECK_HIP(hipMemsetAsync(d_recv, 0, bytes_per_rank * e->tp, s));
        VZ_CHECK_HIP(hipMemcpyAsync((char*)d_recv + bytes_per_rank * e->rank, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, s));
        return VZ_OK;
    }
    if (!e->comm) { vz_set_error("tensor-parallel engine used before vz_comm_init"); return VZ_ERR_STATE; }
    ProfScope ps(e, K_COMM, s);
    ncclResult_t r = ncclAllGather(d_send, d_recv, bytes_per_rank, ncclInt8, e->comm, s);
    if (r != ncclSuccess) { vz_set_error("ncclAllGather failed: %s", ncclGetErrorString(r)); return VZ_ERR_HIP; }
    return VZ_OK;
}

// ------------------------------------------------------------------------------------------------
// a12: Zephyr prefill
// ------------------------------------------------------------------------------------------------
static bf16_t* kc_of(vz_engine* e, int layer) { return e->kv + (size_t)layer * e->kv_layer_elems; }
static bf16_t* vc_of(vz_engine* e, int layer) { return kc_of(e, layer) + e->kv_layer_elems / 2; }

extern "C" int vz_llm_prefill(vz_engine* e, const void* d_embeds, int B, int S, const int* h_seqlens, const int* d_pos,
                              float* d_logits_all, float* d_logits_last, vz_stream stream) {
    return vz_llm_prefill_rows(e, 0, d_embeds, B, S, h_seqlens, d_pos, d_logits_all, d_logits_last, stream);
}

// the same prefill into KV-cache rows row0 .. row0 + B - 1 (continuous batching: a new request enters a free row while the
// other rows keep their context)
extern "C" int vz_llm_prefill_rows(vz_engine* e, int row0, const void* d_embeds, int B, int S, const int* h_seqlens, const int* d_pos,
                                   float* d_logits_all, float* d_logits_last, vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(d_embeds && h_seqlens && d_pos && B >= 1 && row0 >= 0 && row0 + B <= c.max_batch && S >= 1 && S <= c.max_ctx,
                 "prefill: rows %d..%d, S=%d outside capacity (max_batch %d, max_ctx %d)", row0, row0 + B - 1, S, c.max_batch, c.max_ctx);
    const size_t row_off = (size_t)row0 * e->Hkv_l * c.max_ctx * c.head_dim;      // cache rows are [kv head][max_ctx][128] blocks
    VZ_CHECK_ARG(e->cosT && e->rope_max >= c.max_ctx, "prefill: rotary tables not set or shorter than max_ctx");
    for (int b = 0; b < B; ++b) VZ_CHECK_ARG(h_seqlens[b] >= 1 && h_seqlens[b] <= S, "prefill: seqlen[%d]=%d outside [1,%d]", b, h_seqlens[b], S);
    const int H = c.hidden, D = c.head_dim, Hq = e->Hq_l, Hkv = e->Hkv_l, QKV = (Hq + 2 * Hkv) * D, I = e->I_l, A = Hq * D;
    const bool lead = e->rank == 0;   // the row-parallel partial sums carry the residual on one rank only
    const int rows = B * S;
    size_t need;
    {
        Carver m(nullptr, ~(size_t)0);
        m.take<bf16_t>((size_t)rows * H); m.take<bf16_t>((size_t)rows * H); m.take<bf16_t>((size_t)rows * QKV); m.take<bf16_t>((size_t)rows * A);
        m.take<bf16_t>((size_t)rows * A); m.take<bf16_t>((size_t)rows * I); m.take<int>(rows + B + 16); m.take<bf16_t>((size_t)B * H * 2);
        m.take<unsigned char>((size_t)rows * std::max(std::max(H, A), I)); m.take<float>(rows);
        need = m.off + 256;
    }
    RC(ensure_arena(e, need));
    Carver m(e->arena, e->arena_bytes);
    bf16_t* x = m.take<bf16_t>((size_t)rows * H);
    bf16_t* y = m.take<bf16_t>((size_t)rows * H);
    bf16_t* qkv = m.take<bf16_t>((size_t)rows * QKV);
    bf16_t* q = m.take<bf16_t>((size_t)rows * A);
    bf16_t* att = m.take<bf16_t>((size_t)rows * A);
    bf16_t* act = m.take<bf16_t>((size_t)rows * I);
    int* d_ints = m.take<int>(rows + B + 16);   // slot[rows] | seqlens[B]
    bf16_t* ylast = m.take<bf16_t>((size_t)B * H * 2);
    unsigned char* q8 = m.take<unsigned char>((size_t)rows * std::max(std::max(H, A), I));     // e4m3 copy of a linear's input rows + their scales
    float* qs = m.take<float>(rows);                                                           // (fp8 MFMA prefill only)
    int* d_slot = d_ints;
    int* d_len = d_ints + rows;
    {
        std::vector<int> h(rows + B);
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < S; ++t) h[b * S + t] = t < h_seqlens[b] ? t : -1;
        for (int b = 0; b < B; ++b) h[rows + b] = h_seqlens[b];
        RC(upload_ints(e, h.data(), h.size(), d_ints, s));
    }
    int rc = VZ_OK;
    // one Zephyr prefill linear: bf16 tile GEMM, or - weight_fp8 engine with vz_engine_prefill_fp8 on - the input rows quantised to e4m3
    // (one power-of-two scale per row) and the product on the fp8 MFMA against the e4m3 weight copy (gemm_fp8.hip)
    // (below ~768 rows the quantiser launches and the shallow grids cost more than the fp8 MFMA saves: 330 rows 20.4 vs 18.6 ms, 1320 rows 37.5 vs 43.6)
    const bool f8 = e->prefill_fp8 && rows >= g_fp8_prefill_min_rows && tp_local(e) && vz_gemm_fp8_ok(rows, QKV, H, H, H) && vz_gemm_fp8_ok(rows, H, I, I, I) && vz_gemm_fp8_ok(rows, H, A, A, A);
    // norm_w != null: Ain is the residual stream and the RMSNorm belongs to this linear (fp8: norm + quantiser in one launch)
    auto plin = [&](const bf16_t* Ain, const float* norm_w, const std::string& wname, int N, int K, void* Cout, int ldc, const bf16_t* res, int act) -> int {
        const bf16_t* W = WB(wname + ".w", (long)N * K);
        if (rc) return rc;
        if (f8) {
            const unsigned char* w8 = W8(wname + ".w8", (long)N * K);
            const float* ws = WS(wname + ".ws", N);
            if (rc) return rc;
            {
                ProfScope ps(e, K_NORM, s);
                if (norm_w) RC(vz_launch_rmsnorm_quant_fp8(Ain, K, norm_w, c.rms_eps, q8, K, qs, rows, K, s));
                else RC(vz_launch_quant_rows_fp8(Ain, K, q8, K, qs, rows, K, s));
            }
            Fp8LinearArgs f;
            f.A8 = q8; f.lda = K; f.ascale = qs; f.W8 = w8; f.ldw = K; f.wscale = ws; f.C = Cout; f.ldc = ldc; f.M = rows; f.N = N; f.K = K;
            f.bias = nullptr; f.residual = res; f.ldr = H; f.act = act; f.out_fp32 = 0;
            ProfScope ps(e, K_GEMM, s);
            return vz_launch_gemm_fp8(f, s);
        }
        if (norm_w) { ProfScope ps(e, K_NORM, s); RC(vz_launch_rmsnorm(Ain, K, y, K, norm_w, rows, K, c.rms_eps, s)); }
        return linear(e, 0, norm_w ? y : Ain, K, W, K, Cout, ldc, rows, N, K, nullptr, res, H, act, 0, s);
    };
    { ProfScope ps(e, K_OTHER, s); RC(vz_launch_copy_rows((const bf16_t*)d_embeds, H, x, H, rows, H, s)); }
    for (int i = 0; i < c.n_layers; ++i) {
        const std::string p = "llm." + std::to_string(i) + ".";
        RC(plin(x, WF(p + "in_norm", H), p + "qkv", QKV, H, qkv, QKV, nullptr, VZ_ACT_NONE));
        // RoPE of K + the KV append; the queries are rotated by the attention's own Q load (g_rope_in_attn; else a rotated copy q as before)
        const bool rope_q_late = g_rope_in_attn && D == 128 && vz_attn_version() != 1;
        { ProfScope ps(e, K_OTHER, s); RC(vz_launch_rope_kv(qkv, QKV, rope_q_late ? nullptr : q, kc_of(e, i) + row_off, vc_of(e, i) + row_off, e->cosT, e->sinT, d_pos, d_slot, B, S, Hq, Hkv, D, c.max_ctx, s)); }
        {
            ProfScope ps(e, K_ATTN, s);
            AttnArgs a;
            a.q = rope_q_late ? qkv : q; a.k = kc_of(e, i) + row_off; a.v = vc_of(e, i) + row_off; a.o = att;
            a.B = B; a.Sq = S; a.Sk = S; a.Hq = Hq; a.Hkv = Hkv; a.head_dim = D;
            a.q_bs = rope_q_late ? (long)S * QKV : (long)S * A; a.q_ss = rope_q_late ? QKV : A; a.q_hs = D;
            if (rope_q_late) { a.rope_cos = e->cosT; a.rope_sin = e->sinT; a.rope_pos = d_pos; }
            a.k_bs = a.v_bs = (long)Hkv * c.max_ctx * D; a.k_ss = a.v_ss = D; a.k_hs = a.v_hs = (long)c.max_ctx * D;
            a.o_bs = (long)S * A; a.o_ss = A; a.o_hs = D;
            a.scale = 0.08838834764831845f;  // 128^-0.5
            a.causal = 1; a.q_pos0 = 0; a.window = c.sliding_window; a.kv_len = d_len;
            RC(vz_launch_attention(a, s));
        }
        RC(plin(att, nullptr, p + "o", H, A, x, H, lead ? x : nullptr, VZ_ACT_NONE));
        RC(tp_allreduce_bf16(e, x, (size_t)rows * H, s));
        RC(plin(x, WF(p + "post_norm", H), p + "gu", 2 * I, H, act, I, nullptr, VZ_ACT_SWIGLU));
        RC(plin(act, nullptr, p + "down", H, I, x, H, lead ? x : nullptr, VZ_ACT_NONE));
        RC(tp_allreduce_bf16(e, x, (size_t)rows * H, s));
        if (rc) return rc;
    }
    const float* fn = WF("llm.norm", H);
    if (rc) return rc;
    if (d_logits_all) {
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_rmsnorm(x, H, y, H, fn, rows, H, c.rms_eps, s)); }
        RC(lm_head_logits(e, y, rows, d_logits_all, s, nullptr));
    }
    if (d_logits_last) {
        {
            ProfScope ps(e, K_OTHER, s);
            for (int b = 0; b < B; ++b) RC(vz_launch_copy_rows(x + ((size_t)b * S + h_seqlens[b] - 1) * H, H, ylast + (size_t)b * H, H, 1, H, s));
        }
        { ProfScope ps(e, K_NORM, s); RC(vz_launch_rmsnorm(ylast, H, ylast + (size_t)B * H, H, fn, B, H, c.rms_eps, s)); }
        RC(lm_head_logits(e, ylast + (size_t)B * H, B, d_logits_last, s, nullptr));
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// a13: greedy decode
// ------------------------------------------------------------------------------------------------
extern "C" int vz_llm_decode_begin(vz_engine* e, int B, const int* d_first_ids, const int* h_next_pos, const int* h_ctx_len,
                                   vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(B >= 1 && B <= c.max_batch && B <= 64 && d_first_ids && h_next_pos && h_ctx_len, "decode_begin: B=%d unsupported (1..min(64,max_batch))", B);
    const int mb = c.max_batch;
    std::vector<int> h(3 * mb + 4, 0);
    for (int b = 0; b < B; ++b) {
        VZ_CHECK_ARG(h_ctx_len[b] >= 0 && h_ctx_len[b] < c.max_ctx, "decode_begin: ctx_len[%d]=%d outside [0,%d)", b, h_ctx_len[b], c.max_ctx);
        h[b] = h_next_pos[b];            // pos
        h[mb + b] = h_ctx_len[b];        // slot the next token is written to
        h[2 * mb + b] = h_ctx_len[b] + 1;  // keys visible to the next token
    }
    h[3 * mb + 1] = e->samp_ctr0; h[3 * mb + 2] = (int)e->samp_seed[0]; h[3 * mb + 3] = (int)e->samp_seed[1];
    RC(upload_ints(e, h.data(), h.size(), e->d_state + mb, s));   // [pos | slot | len | step = 0, draw counter, seed lo, seed hi]
    VZ_CHECK_HIP(hipMemcpyAsync(e->d_state, d_first_ids, B * sizeof(int), hipMemcpyDeviceToDevice, s));
    e->dec_B = B;
    e->h_len.assign(B, 0); e->h_pos.assign(B, 0); e->h_parked.assign(B, 0);
    for (int b = 0; b < B; ++b) { e->h_len[b] = h_ctx_len[b] + 1; e->h_pos[b] = h_next_pos[b]; e->h_parked[b] = h_ctx_len[b] == 0 && h_next_pos[b] == 0; }
    return VZ_OK;
}

// Continuous batching: (re)arm ONE row of a running decode batch - its next input token, rotary position and context length -
// without touching the other rows or the step counter.  A finished row is parked the same way (any token, position 0, context
// 0): it keeps stepping harmlessly inside its own cache row until a new request is prefilled into it (vz_llm_prefill_rows).
extern "C" int vz_llm_decode_set_row(vz_engine* e, int row, int token, int next_pos, int ctx_len, vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    VZ_CHECK_ARG(e->dec_B >= 1 && row >= 0 && row < e->dec_B, "decode_set_row: row %d outside the running batch of %d", row, e->dec_B);
    VZ_CHECK_ARG(ctx_len >= 0 && ctx_len < c.max_ctx && next_pos >= 0, "decode_set_row: ctx_len %d / pos %d outside [0,%d)", ctx_len, next_pos, c.max_ctx);
    const int mb = c.max_batch;
    const int h[4] = {token, next_pos, ctx_len, ctx_len + 1};     // cur | pos | slot | len: one int in each of the four state arrays
    e->h_len[row] = ctx_len + 1; e->h_pos[row] = next_pos; e->h_parked[row] = ctx_len == 0 && next_pos == 0;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(e->h_pinned && e->h_pinned_ints >= 4, "decode_set_row: no staging buffer (vz_llm_decode_begin allocates it)");
    VZ_CHECK_HIP(hipStreamSynchronize(s));                        // previous use of the staging buffer has drained
    memcpy(e->h_pinned, h, sizeof(h));
    // a 4-row x 4-byte strided copy: row k lands in state array k at column `row`
    VZ_CHECK_HIP(hipMemcpy2DAsync(e->d_state + row, (size_t)mb * sizeof(int), e->h_pinned, sizeof(int), sizeof(int), 4, hipMemcpyHostToDevice, s));
    return VZ_OK;
}

// Batched admissions: several requests are prefilled TOGETHER into spare cache rows (rows the running decode batch does not use,
// vz_llm_prefill_rows with row0 >= the batch size) and then moved to whichever rows have come free - the first h_len[i] cache
// positions of row h_src[i] to row h_dst[i], all layers, one launch per 16 moves, stream-ordered behind the prefill.
extern "C" int vz_llm_kv_move_rows(vz_engine* e, int n, const int* h_src, const int* h_dst, const int* h_len, vz_stream stream) {
    NEED_READY();
    VZ_CHECK_ARG(n >= 1 && h_src && h_dst && h_len, "kv_move_rows: bad argument");
    const vz_config& c = e->c;
    for (int i = 0; i < n; ++i) {
        VZ_CHECK_ARG(h_src[i] >= 0 && h_src[i] < c.max_batch && h_dst[i] >= 0 && h_dst[i] < c.max_batch && h_len[i] >= 0 && h_len[i] <= c.max_ctx,
                     "kv_move_rows: move %d (row %d -> row %d, %d positions) outside the cache (max_batch %d, max_ctx %d)", i, h_src[i], h_dst[i], h_len[i], c.max_batch, c.max_ctx);
        // a row of the running decode batch may only be read or overwritten while it is parked: moving into a live row would replace
        // the keys that row keeps attending to
        VZ_CHECK_ARG(h_src[i] >= e->dec_B || e->h_parked[h_src[i]], "kv_move_rows: source row %d belongs to the running decode batch", h_src[i]);
        VZ_CHECK_ARG(h_dst[i] >= e->dec_B || e->h_parked[h_dst[i]], "kv_move_rows: destination row %d is a live row of the running decode batch (park it first)", h_dst[i]);
    }
    for (int i0 = 0; i0 < n; i0 += 16) {
        KvMoves mv;
        mv.n = std::min(16, n - i0);
        for (int i = 0; i < mv.n; ++i) { mv.src[i] = h_src[i0 + i]; mv.dst[i] = h_dst[i0 + i]; mv.len[i] = h_len[i0 + i]; }
        int r = vz_launch_kv_move_rows(e->kv, e->kv_layer_elems, c.n_layers, c.max_batch, e->Hkv_l, c.max_ctx, c.head_dim, mv, (hipStream_t)stream);
        if (r) return r;
    }
    return VZ_OK;
}

// one decode step, all launches on `s`; every quantity that changes between steps lives in device memory
static int decode_step_launch(vz_engine* e, int* d_out_ids, int out_stride, float* d_logits_dbg, hipStream_t s) {
    const vz_config& c = e->c;
    const int B = e->dec_B, mb = c.max_batch;
    const int H = c.hidden, D = c.head_dim, Hq = e->Hq_l, Hkv = e->Hkv_l, QKV = (Hq + 2 * Hkv) * D, I = e->I_l, A = Hq * D;
    const bool lead = e->rank == 0;
    int* cur = e->d_state; int* pos = cur + mb; int* slot = pos + mb; int* len = slot + mb; int* step = len + mb;
    Carver m(e->arena, e->arena_bytes);
    bf16_t* x = m.take<bf16_t>((size_t)B * H);
    bf16_t* qkv = m.take<bf16_t>((size_t)B * QKV);
    bf16_t* att = m.take<bf16_t>((size_t)B * A);
    bf16_t* act = m.take<bf16_t>((size_t)B * I);
    if (!m.ok) { vz_set_error("decode: workspace too small"); return VZ_ERR_STATE; }
    int rc = VZ_OK;
    if (e->use_tok && e->tok) {
        // batch 1 on an MI355X: embedding row -> 32 layers -> logits as the phases of ONE resident grid (decode_persist.hip); same
        // arithmetic as the launches below, bit for bit
        VzTokArgs a;
        a.embed = WB("llm.embed", (long)c.vocab * H); a.lm_head = WB("llm.lm_head", (long)e->Vp * H); a.final_norm = WF("llm.norm", H);
        if (rc) return rc;
        a.cur = cur; a.pos = pos; a.slot = slot; a.step = step;
        a.logits = e->d_logits; a.part = e->d_part; a.ticket = e->d_ticket; a.cosT = e->cosT; a.sinT = e->sinT; a.err = e->d_ferr;
        a.vocab = c.vocab; a.max_ctx = c.max_ctx; a.nsplit = e->dec_nsplit; a.window = c.sliding_window; a.scale = 0.08838834764831845f; a.eps = c.rms_eps;
        e->last_stream = s;
        ProfScope ps(e, K_GEMV, s);
        RC(vz_launch_decode_token(e->tok, a, s));
    } else {
        { ProfScope ps(e, K_OTHER, s); RC(vz_launch_embed_tokens(cur, B, H, WB("llm.embed", (long)c.vocab * H), x, s)); }
        for (int i = 0; i < c.n_layers; ++i) {
            const std::string p = "llm." + std::to_string(i) + ".";
            const bool fuse_ao = g_attn_o && (B == 1 || (B == 2 && g_attn_o >= 1 && 2 * e->dec_nsplit <= 32)) && e->tp == 1 && tp_local(e) && H == 4096 && A == 4096 && Hq == 32 && Hkv == 8 &&
                                 D == 128 && e->dec_nsplit <= 32 && e->d_ao_done;
            RC(linear(e, 1, x, H, WB(p + "qkv.w", (long)QKV * H), H, qkv, QKV, B, QKV, H, nullptr, nullptr, 0, VZ_ACT_NONE, 0, s, WF(p + "in_norm", H), c.rms_eps,
                      W8(p + "qkv.w8", (long)QKV * H), WS(p + "qkv.ws", QKV)));
            {
                ProfScope ps(e, K_ATTN_DEC, s);
                AttnDecodeFusedArgs a;
                a.qkv = qkv; a.kc = kc_of(e, i); a.vc = vc_of(e, i); a.o = att; a.part = e->d_part; a.ticket = e->d_ticket;
                a.cosT = e->cosT; a.sinT = e->sinT; a.pos = pos; a.slot = slot;
                a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.D = D; a.max_ctx = c.max_ctx; a.nsplit = e->dec_nsplit; a.window = c.sliding_window;
                a.scale = 0.08838834764831845f;
                if (fuse_ao) {
                    // batch 1 (round 3): the O projection's workgroups ride in the attention's grid and stream their weights under its latency
                    // chain (attn_o_fused.hip); same arithmetic as the two launches, bit for bit
                    const bf16_t* ow = WB(p + "o.w", (long)H * A);
                    const unsigned char* ow8 = W8(p + "o.w8", (long)H * A);        // a weight_fp8 engine: the e4m3 rows + scales (as linear() would take them)
                    const float* ows = WS(p + "o.ws", H);
                    if (rc) return rc;
                    RC(vz_launch_attn_o_fused(a, ow, ow8, ows, att, x, e->d_ao_done, step, i, c.n_layers, e->d_ferr, s));
                } else {
                    RC(vz_launch_attn_decode_fused(a, s));
                }
            }
            if (!fuse_ao)
                RC(linear(e, 1, att, A, WB(p + "o.w", (long)H * A), A, x, H, B, H, A, nullptr, lead ? x : nullptr, H, VZ_ACT_NONE, 0, s, nullptr, 0.f,
                          W8(p + "o.w8", (long)H * A), WS(p + "o.ws", H)));
            RC(tp_allreduce_bf16(e, x, (size_t)B * H, s, true));
            RC(linear(e, 1, x, H, WB(p + "gu.w", 2L * I * H), H, act, I, B, 2 * I, H, nullptr, nullptr, 0, VZ_ACT_SWIGLU, 0, s, WF(p + "post_norm", H), c.rms_eps,
                      W8(p + "gu.w8", 2L * I * H), WS(p + "gu.ws", 2L * I)));
            RC(linear(e, 1, act, I, WB(p + "down.w", (long)I * H), I, x, H, B, H, I, nullptr, lead ? x : nullptr, H, VZ_ACT_NONE, 0, s, nullptr, 0.f,
                      W8(p + "down.w8", (long)I * H), WS(p + "down.ws", H)));
            RC(tp_allreduce_bf16(e, x, (size_t)B * H, s, true));
            if (rc) return rc;
        }
        RC(lm_head_logits(e, x, B, e->d_logits, s, WF("llm.norm", H)));
    }
    if (rc) return rc;
    if (d_logits_dbg) {
        // debug copy is indexed by the host (eager mode only)
        VZ_CHECK_HIP(hipMemcpyAsync(d_logits_dbg, e->d_logits, (size_t)B * c.vocab * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    {
        ProfScope ps(e, K_OTHER, s);
        if (e->samp_on)
            RC(vz_launch_sample(e->d_logits, B, c.vocab, e->samp_temp, e->samp_top_k, e->samp_top_p, (const unsigned*)(step + 2), step + 1, 0, cur, pos,
                                slot, len, d_out_ids, out_stride, step, c.max_ctx, e->rope_max, e->ring, e->ring_n, s));
        else
            RC(vz_launch_argmax(e->d_logits, B, c.vocab, cur, pos, slot, len, d_out_ids, out_stride, step, c.max_ctx, e->rope_max, e->ring, e->ring_n, s));
        RC(vz_launch_step_advance(step, s));
    }
    return VZ_OK;
}

// Sampling instead of argmax as the tail of every decode step (hf:generation/utils.py `_sample`, do_sample=True; see
// sampling.hip): temperature > 0, top_k (0 = off; HF's default 50 is the caller's business), top_p (1 = off), a 64-bit seed and
// the draw counter the NEXT vz_llm_decode_begin starts from (the caller drew token 0 from the prefill logits with vz_op_sample
// and counter 0, so it passes 1).  enable = 0: greedy.
extern "C" int vz_llm_decode_sampling(vz_engine* e, int enable, float temperature, int top_k, float top_p, unsigned long long seed,
                                      int first_counter) {
    VZ_CHECK_ARG(e && first_counter >= 0, "decode_sampling: null engine / negative counter");
    e->samp_ctr0 = first_counter;        // the draw counter also indexes the host-visible token ring of a greedy streamer loop
    if (!enable) { e->samp_on = 0; return VZ_OK; }
    VZ_CHECK_ARG(temperature > 0.f && top_k >= 0 && top_p > 0.f && top_p <= 1.f && first_counter >= 0,
                 "decode_sampling: temperature %g > 0, top_k %d >= 0, 0 < top_p %g <= 1 expected", (double)temperature, top_k, (double)top_p);
    e->samp_on = 1; e->samp_temp = temperature; e->samp_top_k = top_k; e->samp_top_p = top_p;
    e->samp_seed[0] = (unsigned)seed; e->samp_seed[1] = (unsigned)(seed >> 32); e->samp_ctr0 = first_counter;
    return VZ_OK;
}

// Streamer / stopping-criteria path: besides d_out_ids every step's tail also writes its token to ring[row * ring_n + (draw
// counter mod ring_n)], a DEVICE-VISIBLE HOST buffer (hipHostMalloc / pinned), so the host can keep a step or two in flight and
// read token t as soon as the event recorded behind step t fires, without a device-to-host copy per token.  NULL = off.
extern "C" int vz_llm_decode_ring(vz_engine* e, int* ring, int ring_n, int ring_rows) {
    VZ_CHECK_ARG(e && (!ring || (ring_n >= 2 && ring_rows >= 1)), "decode_ring: ring_n >= 2 slots and ring_rows >= 1 rows expected");
    e->ring = ring; e->ring_n = ring ? ring_n : 0; e->ring_rows = ring ? ring_rows : 0;
    return VZ_OK;
}

// one draw per row of fp32 logits [rows, cols] with the same kernel (the first token of a sampled generation; tests)
extern "C" int vz_op_sample(const float* d_logits, int rows, int cols, float temperature, int top_k, float top_p,
                            unsigned long long seed, int counter, int* d_ids, vz_stream stream) {
    // [counter, seed lo, seed hi, -] in this (device, stream)'s own scratch words: launches on other streams or devices have theirs
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(d_logits && d_ids && rows >= 1 && cols >= 1, "sample: bad argument");
    int* d_scratch = nullptr;
    { void* p = nullptr; size_t have = 0; int r = vz_stream_ws(4, s, 64, true, &p, &have); if (r) return r;
      VZ_CHECK_ARG(p && have >= 16, "sample: first use of a stream inside a capture (call it once before capturing)"); d_scratch = (int*)p; }
    VZ_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)d_scratch, counter, 1, s));
    VZ_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(d_scratch + 1), (int)(unsigned)seed, 1, s));
    VZ_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)(d_scratch + 2), (int)(unsigned)(seed >> 32), 1, s));
    return vz_launch_sample(d_logits, rows, cols, temperature, top_k, top_p, (const unsigned*)(d_scratch + 1), d_scratch, 0, d_ids, nullptr,
                            nullptr, nullptr, nullptr, 0, nullptr, 0, 0, nullptr, 0, s);
}

extern "C" int vz_llm_decode_steps(vz_engine* e, int n, int* d_out_ids, float* d_logits_dbg, vz_stream stream) {
    NEED_READY();
    const vz_config& c = e->c;
    hipStream_t s = (hipStream_t)stream;
    VZ_CHECK_ARG(e->dec_B >= 1, "decode_steps: call vz_llm_decode_begin first");
    VZ_CHECK_ARG(n >= 1 && d_out_ids, "decode_steps: bad argument");
    const int B = e->dec_B;
    VZ_CHECK_ARG(!e->ring || B <= e->ring_rows, "decode_steps: the token ring holds %d rows, the decode batch has %d", e->ring_rows, B);
    const size_t need = ((size_t)B * (3 * c.hidden + (c.n_heads + 2 * c.n_kv_heads) * c.head_dim + c.inter)) * 2 + 8192;   // upper bound (tp = 1 sizes)
    RC(ensure_arena(e, need));
    // tensor-parallel steps: the RCCL all-reduces / all-gather are captured with the kernels (one graph launch per token instead of
    // ~230 host launches); if RCCL refuses the capture the engine falls back to eager steps for good
    bool use_graph = !e->prof_on && !d_logits_dbg && getenv("VZ_NO_GRAPH") == nullptr &&
                     (tp_local(e) || (e->comm_graph_ok && getenv("VZ_TP_NO_GRAPH") == nullptr));
    if (!tp_local(e)) RC(ensure_gather(e, B, s));
    int* step = e->d_state + 4 * c.max_batch;
    VZ_CHECK_HIP(hipMemsetAsync(step, 0, sizeof(int), s));
    VZ_CHECK_HIP(hipMemsetAsync(e->d_ao_done, 0, sizeof(unsigned), s));      // the attention + O launch's arrival word restarts with the step counter
    // vz_tune_set(28, 1) - batch 1, one GPU, bf16 weights, Zephyr-7B geometry on a 256-CU device: the steps run as ONE resident grid
    // per token (decode_persist.hip) instead of the launch chain.  Its pointer table is (re)built here, never inside a capture.
    e->use_tok = false;
    if (g_persist_decode && B == 1 && e->tp == 1 && tp_local(e) && !c.weight_fp8 && c.hidden == 4096 && c.inter == 14336 && c.n_heads == 32 &&
        c.n_kv_heads == 8 && c.head_dim == 128 && c.vocab >= 256 && vz_decode_persist_supported()) {
        if (!e->tok) {
            std::vector<VzTokLayerHost> lt(c.n_layers);
            int rc = VZ_OK;
            const long H = c.hidden, QKV = (long)(c.n_heads + 2 * c.n_kv_heads) * c.head_dim, I = c.inter;
            for (int i = 0; i < c.n_layers; ++i) {
                const std::string p = "llm." + std::to_string(i) + ".";
                lt[i].qkv_w = WB(p + "qkv.w", QKV * H); lt[i].o_w = WB(p + "o.w", H * H); lt[i].gu_w = WB(p + "gu.w", 2 * I * H); lt[i].down_w = WB(p + "down.w", I * H);
                lt[i].in_norm = WF(p + "in_norm", H); lt[i].post_norm = WF(p + "post_norm", H);
                lt[i].kc = kc_of(e, i); lt[i].vc = vc_of(e, i);
            }
            if (rc) return rc;
            RC(vz_decode_persist_create(lt.data(), c.n_layers, &e->tok));
        }
        RC(vz_decode_persist_reset(e->tok, s));         // arrival counters restart with the step counter
        if (e->tok_poke_word >= 0) { RC(vz_decode_persist_poke(e->tok, e->tok_poke_word, e->tok_poke_value, s)); e->tok_poke_word = -1; }
        e->use_tok = true;
    }
    // Capacity (the cache append writes slot = len - 1 of the row, the rotary tables are read at pos): every live row must still fit
    // after n steps.  Parked rows (continuous batching) are not checked: the step tail saturates their slot / position on the device.
    int len_max = 0;
    for (int b = 0; b < B; ++b) {
        if (!e->h_parked[b]) {
            VZ_CHECK_ARG(e->h_len[b] + n - 1 <= c.max_ctx, "decode_steps: row %d would reach %d keys, the cache holds max_ctx = %d", b, e->h_len[b] + n - 1, c.max_ctx);
            VZ_CHECK_ARG(e->h_pos[b] + n - 1 < e->rope_max, "decode_steps: row %d would reach position %d, the rotary tables hold %d", b, e->h_pos[b] + n - 1, e->rope_max);
        }
        len_max = std::max(len_max, std::min(e->h_len[b], c.max_ctx));
    }
    e->dec_len_max = len_max;
    // Context splits of the decode attention = grid.x: a split takes >= 128 keys and workgroups that find nothing to do still cost
    // a dispatch slot each (measured: 4096 mostly idle workgroups = 59 us per layer at 16 rows), so the grid follows the longest
    // context these n steps can reach - known on the host - in coarse buckets (a new bucket = one re-capture of the graph).
    {
        int keys = std::min(e->dec_len_max + n, c.max_ctx);
        if (c.sliding_window > 0) keys = std::min(keys, c.sliding_window);
        const int need = (keys + 127) / 128;
        static const int buckets[] = {1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32};
        int ns = e->nsplit;
        for (int bk : buckets) if (bk >= need) { ns = std::min(bk, e->nsplit); break; }
        // many rows fill the chip by themselves: splitting their contexts as well only adds dispatch slots and a merge hop per (row, KV head)
        // (config-5 miniature, e4m3, ctx ~330: 64 rows x 3 splits -> x 1: decode 627 -> 597 ms per batch; 32 rows x 3 -> x 2: 459 -> 448) -
        // rows x KV heads x splits is held to ~512 workgroups; 1..2 rows keep every split (batch-1 numbers unchanged)
        ns = std::min(ns, std::max(1, 512 / std::max(1, B * e->Hkv_l)));
        e->dec_nsplit = g_attn_nsplit > 0 ? g_attn_nsplit : ns;
        for (int b = 0; b < B; ++b) {       // what the device-side state will be after these n steps (the tail saturates, so do we)
            if (e->h_parked[b]) { e->h_len[b] = std::min(e->h_len[b] + n, c.max_ctx); e->h_pos[b] = std::min(e->h_pos[b] + n, e->rope_max - 1); }
            else { e->h_len[b] += n; e->h_pos[b] += n; }      // a live row that is full is refused by the check above on the next call
        }
    }
    if (!use_graph) {
        for (int i = 0; i < n; ++i)
            RC(decode_step_launch(e, d_out_ids, n, d_logits_dbg ? d_logits_dbg + (size_t)i * B * c.vocab : nullptr, s));
        return VZ_OK;
    }
    // Output pointer / stride, the workspace and the sampling parameters are kernel arguments frozen in the graph: re-capture when
    // they change (seed and draw counter live in device memory and do not).
    long samp_key[6] = {e->samp_on, e->samp_top_k, 0, 0, (long)(uintptr_t)e->ring, e->ring_n};
    memcpy(&samp_key[2], &e->samp_temp, 4); memcpy(&samp_key[3], &e->samp_top_p, 4);
    if (!e->dec_graph || e->dec_graph_B != B || e->dec_graph_n != n || e->dec_graph_out != d_out_ids || e->dec_graph_arena != e->arena ||
        e->dec_graph_nsplit != e->dec_nsplit || e->dec_graph_tok != ((int)e->use_tok | (g_attn_o << 1)) || memcmp(e->dec_graph_samp, samp_key, sizeof(samp_key)) != 0) {
        if (e->dec_graph) { hipGraphExecDestroy(e->dec_graph); e->dec_graph = nullptr; }
        hipGraph_t graph;
        if (!e->cap_stream) VZ_CHECK_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
        if (B >= 17) {     // split-K scratch of the capture stream (128^2 tile route, gemm_wide K splits): never allocated inside a capture
            void* p = nullptr; size_t have = 0;
            RC(vz_stream_ws(0, e->cap_stream, (size_t)96 << 20, false, &p, &have));
            RC(vz_wide_reserve(e->cap_stream));
        }
        VZ_CHECK_HIP(hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
        int r = decode_step_launch(e, d_out_ids, n, nullptr, e->cap_stream);
        graph = nullptr;
        hipError_t er = hipStreamEndCapture(e->cap_stream, &graph);
        if (r == VZ_OK && er == hipSuccess) er = hipGraphInstantiate(&e->dec_graph, graph, nullptr, nullptr, 0);
        if (graph) hipGraphDestroy(graph);
        if (r != VZ_OK || er != hipSuccess) {
            e->dec_graph = nullptr;
            if (tp_local(e)) { if (r) return r; VZ_CHECK_HIP(er); }
            (void)hipGetLastError();            // collectives would not capture: eager steps from now on
            e->comm_graph_ok = false;
            for (int i = 0; i < n; ++i) RC(decode_step_launch(e, d_out_ids, n, nullptr, s));
            return VZ_OK;
        }
        e->dec_graph_B = B; e->dec_graph_n = n; e->dec_graph_out = d_out_ids; e->dec_graph_arena = e->arena; e->dec_graph_nsplit = e->dec_nsplit; e->dec_graph_tok = (int)e->use_tok | (g_attn_o << 1); memcpy(e->dec_graph_samp, samp_key, sizeof(samp_key));
    }
    for (int i = 0; i < n; ++i) VZ_CHECK_HIP(hipGraphLaunch(e->dec_graph, s));
    return VZ_OK;
}

// weight_fp8 engines: 1 = the Zephyr prefill linears quantise their input rows to e4m3 and run on the fp8 MFMA against the e4m3 weight
// copies (config 5's "fp8 MFMA weights"; gemm_fp8.hip); 0 (default) = bf16 MFMA on the dequantised bf16 tensors.  Decode is untouched.
extern "C" int vz_engine_prefill_fp8(vz_engine* e, int enable) {
    VZ_CHECK_ARG(e && (!enable || e->c.weight_fp8), "prefill_fp8: needs an engine created with weight_fp8 = 1");
    e->prefill_fp8 = enable ? 1 : 0;
    return VZ_OK;
}

// how the last vz_llm_decode_steps ran: *graph = 1 if a captured graph was replayed, *comm_in_graph = 1 if the engine's RCCL
// collectives are part of it (tensor-parallel engines; 0 after a refused capture = eager steps)
extern "C" int vz_llm_decode_mode(vz_engine* e, int* graph, int* comm_in_graph) {
    VZ_CHECK_ARG(e && graph && comm_in_graph, "decode_mode: null argument");
    *graph = e->dec_graph != nullptr;
    *comm_in_graph = e->dec_graph != nullptr && !tp_local(e) && e->comm_graph_ok;
    return VZ_OK;
}

// Device-side waits of the one-launch attention half are bounded; one that expires raises a word the host reads here (blocking
// 4-byte copy after a stream sync; the word is cleared).  *err != 0 means the ids / logits of the steps since the last call are garbage.
extern "C" int vz_engine_async_error(vz_engine* e, int* err) {
    VZ_CHECK_ARG(e && err, "async_error: null argument");
    *err = 0;
    if (!e->d_ferr) return VZ_OK;
    VZ_CHECK_HIP(hipMemcpy(err, e->d_ferr, sizeof(int), hipMemcpyDeviceToHost));
    if (*err) {
        VZ_CHECK_HIP(hipMemset(e->d_ferr, 0, sizeof(int)));
        // the stream-K tickets that made a wait expire are in an unknown state: start the next launch from zero
        int dummy = 0;
        RC(vz_gemm256_async_error(e->last_stream, &dummy, true));
    }
    return VZ_OK;
}

// the same word for op-level launches on `stream` (vz_op_linear* taking the stream-K path): blocking read + clear
extern "C" int vz_op_async_error(vz_stream stream, int* err) {
    VZ_CHECK_ARG(err, "op_async_error: null argument");
    return vz_gemm256_async_error((hipStream_t)stream, err, false);
}

// TEST HOOK: overwrite the {arrive, ready} pair of stream-K remainder tile `tile` on `stream` (tests/test_ops_gpu.py drives an
// expired fix-up wait with it: the launch must end, raise VZ_ASYNC_STREAMK and write NaN, never a sum of stale slots)
// TEST HOOK: preset arrival-counter shard `word` (0..7; 8 = the abort word) of the persistent decode-token kernel's hand-off state for
// the NEXT vz_llm_decode_steps call - tests/test_persist_gpu.py makes a shard lag so that a wait can never be met: the launch must END,
// raise VZ_ASYNC_PERSIST, and the call after it (which zeroes the counters again) must be clean.  *mode (may be null) receives 1 if the last steps ran on the
// persistent kernel.
extern "C" int vz_test_persist_poke(vz_engine* e, int word, unsigned value, int* mode, vz_stream stream) {
    VZ_CHECK_ARG(e && word <= 8, "persist_poke: bad argument");
    (void)stream;
    if (mode) *mode = e->use_tok && e->tok ? 1 : 0;
    if (word >= 0) { e->tok_poke_word = word; e->tok_poke_value = value; }      // applied by the next vz_llm_decode_steps, behind its counter reset
    return VZ_OK;
}
extern "C" int vz_test_corrupt_streamk(vz_stream stream, int tile, int arrive, int ready) {
    return vz_gemm256_corrupt_tickets((hipStream_t)stream, tile, arrive, ready);
}

extern int g_attn_o_delay;
extern int g_skinny_even;
extern int g_gemm256_streamk, g_gemm256_skew, g_gemm256_stamps, g_gemm256_drain, g_gemm256_persist, g_attn_stamp_on, g_fp8_gemm_choice;
int vz_gemm256_read_stamps(long long* host, int max_wgs, int* n_wgs);
extern "C" int vz_tune_set(int knob, int value) {
    if (knob == 0) { vz_set_gemv_variant(value); return VZ_OK; }
    if (knob == 1) { vz_set_gemm_choice(value); return VZ_OK; }
    if (knob == 2) { vz_set_attn_version(value); return VZ_OK; }
    if (knob == 3) { vz_set_splitk_mode(value); return VZ_OK; }
    if (knob == 4) { g_gemm256_streamk = value; return VZ_OK; }
    if (knob == 5) { g_gemm256_skew = value; return VZ_OK; }
    if (knob == 6) { g_gemm256_stamps = value; return VZ_OK; }
    if (knob == 7) { g_force_comm = value; return VZ_OK; }
    if (knob == 9) { g_skinny_mode = value; return VZ_OK; }
    if (knob == 11) { g_gemm256_drain = value; return VZ_OK; }
    if (knob == 14) { g_decode_tile_rows = value; return VZ_OK; }
    if (knob == 19) { g_wide_mode = value; return VZ_OK; }
    if (knob == 21) { g_fp8_gemm_choice = value; return VZ_OK; }
    if (knob == 22) { g_fp8_prefill_min_rows = value; return VZ_OK; }
    if (knob == 23) { vz_set_attn_split(value); return VZ_OK; }
    if (knob == 24) { vz_set_splitk_cap(value); return VZ_OK; }
    if (knob == 25) { g_qf_kv_all = value; return VZ_OK; }
    if (knob == 27) { g_wide_fp8_splits = value; return VZ_OK; }
    if (knob == 28) { g_persist_decode = value; return VZ_OK; }
    if (knob == 29) { g_oneshot = value; return VZ_OK; }
    if (knob == 30) { g_attn_o = value; return VZ_OK; }
    if (knob == 31) { g_attn_o_delay = value; return VZ_OK; }
    if (knob == 32) { g_flash_bwd = value; return VZ_OK; }
    if (knob == 33) { g_rope_in_attn = value; return VZ_OK; }
    if (knob == 34) { g_gemm256_persist = value; return VZ_OK; }
    if (knob == 35) { g_skinny_even = value; return VZ_OK; }
    if (knob == 26) { vz_set_splitk_mid(value); return VZ_OK; }
    if (knob == 15) { g_decode_sk_short = value; return VZ_OK; }
    if (knob == 16) { g_attn_stamp_on = value; return VZ_OK; }
    if (knob == 10) { if (value < 0 || value > 64) { vz_set_error("tune_set: decode attention splits must be 0..64"); return VZ_ERR_ARG; } g_attn_nsplit = value; return VZ_OK; }
    vz_set_error("tune_set: unknown knob %d", knob);
    return VZ_ERR_ARG;
}

int vz_attn_read_stamps(long long* host16);
extern "C" int vz_prof_attn_stamps(long long* host16) { return vz_attn_read_stamps(host16); }
extern "C" int vz_prof_gemm_stamps(long long* host_out, int max_wgs, int* n_wgs) {
    return vz_gemm256_read_stamps(host_out, max_wgs, n_wgs);
}

// phase stamps of the last token the persistent decode-token kernel ran (tools/persist_stamps.py): [n_layers][12] 100 MHz ticks
extern "C" int vz_prof_persist_stamps(vz_engine* e, unsigned long long* host, int n_layers) {
    VZ_CHECK_ARG(e && e->tok, "prof_persist_stamps: the engine has not run a step on the persistent kernel");
    return vz_decode_persist_stamps(e->tok, host, n_layers);
}

extern "C" int vz_prof_enable(vz_engine* e, int enable, int klass) {
    VZ_CHECK_ARG(e, "prof: null engine");
    e->prof_on = enable; e->prof_class = klass; e->prof_used = 0;
    return VZ_OK;
}
extern "C" int vz_prof_read(vz_engine* e, long* n_launches, double* total_ms) {
    VZ_CHECK_ARG(e && n_launches && total_ms, "prof: null argument");
    double tot = 0;
    for (size_t i = 0; i < e->prof_used; ++i) {
        VZ_CHECK_HIP(hipEventSynchronize(e->prof_ev[i].second));
        float ms = 0;
        VZ_CHECK_HIP(hipEventElapsedTime(&ms, e->prof_ev[i].first, e->prof_ev[i].second));
        tot += ms;
    }
    *n_launches = (long)e->prof_used; *total_ms = tot;
    e->prof_used = 0;
    return VZ_OK;
}

#include "train_engine.inc"
