// Shared device helpers and launch plumbing for libviszephyr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/viszephyr.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

#define VZ_WAVE 64

__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

// ---- wave-wide reductions on the VALU (gfx950) ----
// Inside a 16-lane row: DPP moves (quad swaps, half-row mirror, row mirror) - after the four steps every lane of a row holds
// its row's result.  Across the four rows: v_permlane16_swap (odd rows of one operand <-> even rows of the other) and
// v_permlane32_swap (wave halves); with both operands = v the pair is (r0,r0,r2,r2) / (r1,r1,r3,r3) resp. (lo,lo) / (hi,hi), so
// one op() of the pair is the xor-16 resp. xor-32 butterfly step.  Six VALU steps instead of six ds_bpermute round trips
// through the LDS pipeline; the result is valid in EVERY lane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) { return v + dpp_mov<CTRL, ROW_MASK>(v); }

__device__ __forceinline__ float rows_sum(float v) {       // all-reduce over lanes c, c+16, c+32, c+48
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float rows_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xb1, 0xf>(v);      // quad_perm:[1,0,3,2]
    v = dpp_add<0x4e, 0xf>(v);      // quad_perm:[2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror
    return rows_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xb1, 0xf>(v));
    v = fmaxf(v, dpp_mov<0x4e, 0xf>(v));
    v = fmaxf(v, dpp_mov<0x141, 0xf>(v));
    v = fmaxf(v, dpp_mov<0x140, 0xf>(v));
    return rows_max(v);
}
// sum valid in LANE 63 ONLY (row_bcast:15 / row_bcast:31 instead of the two swaps: one instruction less per value when many
// values are reduced at once, as the GEMV epilogue does)
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0xb1, 0xf>(v);
    v = dpp_add<0x4e, 0xf>(v);
    v = dpp_add<0x141, 0xf>(v);
    v = dpp_add<0x140, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 -> rows 2, 3
    return v;
}

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + __expf(-x)); }

// ---- host-side error plumbing ----
void vz_set_error(const char* fmt, ...);
#define VZ_CHECK_ARG(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            vz_set_error(__VA_ARGS__);           \
            return VZ_ERR_ARG;                   \
        }                                        \
    } while (0)
#define VZ_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            vz_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VZ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)
#define VZ_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t _e = hipGetLastError();                                                   \
        if (_e != hipSuccess) {                                                              \
            vz_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return VZ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ---- once-per-DEVICE initialisation (hipFuncSetAttribute is a per-device setting: a process-wide `static bool done` would leave
// the second device of a process with the default dynamic-LDS limit) ----
struct VzDeviceOnce { unsigned long long seen[4] = {0, 0, 0, 0}; };
bool vz_device_first(VzDeviceOnce& o);      // true exactly once per (o, current device); thread-safe (gemm.hip)
// per-(kind, device, stream) scratch for split-K slabs / partial tiles / arrival tickets (gemm.hip).  kind 0: 128^2 GEMM slab,
// 1: gemm_wide partial tiles, 2: gemm_wide tickets (zeroed), 3: decode-chain hand-off state (zeroed).  Inside a stream capture nothing is
// allocated: *out / *out_bytes report what exists (reserve before capturing).
int vz_stream_ws(int kind, hipStream_t s, size_t min_bytes, bool zero, void** out, size_t* out_bytes);

// ---- internal launchers shared between the op-level C ABI and the engine ----
struct LinearArgs {
    const bf16_t* A; int lda;
    const bf16_t* W; int ldw;
    void* C; int ldc;
    int M, N, K;
    const float* bias;
    const bf16_t* residual; int ldr;
    int act; int out_fp32;
    // optional fused RMSNorm prologue (GEMV path only): x <- bf16(norm_w * x * rsqrt(mean(x^2)+eps))
    const float* norm_w; float norm_eps;
    // optional e4m3 copy of W (rows [N][ldw] bytes) + one fp32 power-of-two scale per row: GEMV path streams these instead
    const unsigned char* W8 = nullptr; const float* wscale = nullptr;
    // optional MFMA-fragment-tiled copy of W (vz_launch_tile_weights: [N/16][K/64][2][64 lanes][8]): the 2..64-row weight stream
    // (gemm_skinny.hip) then reads 1 KiB contiguous per wave-instruction instead of 16 rows x 64 bytes; same values, same k order
    const bf16_t* Wt = nullptr;
    // optional fragment-tiled copy of W8 (vz_launch_tile_weights_fp8: [N/16][K/64][64 lanes][16]): the 17..64-row e4m3 stream of gemm_wide.hip
    const unsigned char* W8t = nullptr;
    // 17..64 rows may take the MFMA weight stream (rows = independent sequences of a decode batch).  Off for the engine's prefill /
    // Q-Former linears: there a row's result must not depend on how many rows sit beside it (the tile GEMM's split-K is a
    // function of N and K only; tests/test_stages_gpu.py::test_qformer), and 32 / 64 / 96 rows must all take the same kernel.
    bool wide_ok = true;
    // 128^2 tile GEMM only: split-K factor to use instead of the batch-invariant default (0 = default).  A decode batch of 17..64 rows
    // (independent sequences, weight-stream bound) cuts K finer so that N / 128 column tiles fill the 256 CUs
    int splitk_hint = 0;
    // async error word of the caller (an engine's); nullptr = the launch stream's own (vz_op_async_error)
    int* err = nullptr;
};
int vz_launch_gemm(const LinearArgs& a, hipStream_t s);
int vz_launch_tile_weights(const bf16_t* W, int N, int K, int ldw, bf16_t* Wt, hipStream_t s);
// gemm_fp8.hip: e4m3 x e4m3 MFMA GEMM (per-row power-of-two scales on both operands) + the activation quantiser
struct Fp8LinearArgs {
    const unsigned char* A8; int lda; const float* ascale;      // e4m3 activations [M, lda bytes] + 2^e per row
    const unsigned char* W8; int ldw; const float* wscale;      // e4m3 weights [N, ldw bytes] + 2^e per row
    void* C; int ldc; int M, N, K;
    const float* bias; const bf16_t* residual; int ldr; int act; int out_fp32;
};
int vz_launch_quant_rows_fp8(const bf16_t* x, int ldx, unsigned char* q, int ldq, float* scale, int rows, int K, hipStream_t s);
int vz_launch_rmsnorm_quant_fp8(const bf16_t* x, int ldx, const float* w, float eps, unsigned char* q, int ldq, float* scale, int rows, int cols, hipStream_t s);
bool vz_gemm_fp8_ok(int M, int N, int K, int lda, int ldw);
int vz_launch_gemm_fp8(const Fp8LinearArgs& a, hipStream_t s);
int vz_launch_gemm256_fp8(const Fp8LinearArgs& a, hipStream_t s);      // gemm256.hip's pipeline on e4m3 operands
// gemm_wide.hip: 17..64 rows on the tiled weight copy, activations staged once per 128 weight rows (needs a.Wt, no fused norm)
bool vz_wide_ok(const LinearArgs& a);
bool vz_wide_engine_ok(const LinearArgs& a);     // the shapes an engine's decode step routes there (no K split)
int vz_launch_wide(const LinearArgs& a, hipStream_t s);
int vz_launch_tile_weights_fp8(const unsigned char* W8, int N, int K, int ldw, unsigned char* W8t, hipStream_t s);
int vz_wide_reserve(hipStream_t s);             // K-split scratch of a stream, allocated outside a capture
extern int g_wide_fp8_splits;
int vz_init_wide_kernels();
extern int g_wide_mode;
int vz_linear_check_common(const LinearArgs& a);
bool vz_gemv_ok(const LinearArgs& a);
bool vz_skinny_ok(const LinearArgs& a);      // 2..16 rows: MFMA weight stream (gemm_skinny.hip)
int vz_launch_skinny(const LinearArgs& a, hipStream_t s);
int vz_init_skinny_kernels();
extern int g_skinny_mode;
bool vz_skinny_fused_norm_ok(const LinearArgs& a);
int vz_init_gemv_kernels();
void vz_set_gemv_variant(int v);
// Profiling: when set, the next GEMM/GEMV launch is issued through hipExtLaunchKernelGGL with these events, which
// the runtime stamps at the kernel's own start and end on the GPU (no launch gap inside the bracket).
extern thread_local hipEvent_t g_vz_prof_start, g_vz_prof_stop;
#include <hip/hip_ext.h>
template <typename F, typename... Args>
inline void vz_launch_timed(F kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, Args... args) {
    if (g_vz_prof_start) {
        hipExtLaunchKernelGGL(kernel, grid, block, lds, s, g_vz_prof_start, g_vz_prof_stop, 0, args...);
        g_vz_prof_start = nullptr;
        g_vz_prof_stop = nullptr;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
    }
}
struct BatchedGemmArgs {
    const bf16_t* A; int lda; const bf16_t* W; int ldw; void* C; int ldc;
    int M, N, K, out_fp32;
    int n_outer, n_inner, a_div, w_div;
    long a_so, a_si, w_so, w_si, c_so, c_si;       // element strides of the outer / inner batch index per operand
};
int vz_launch_gemm_batched(const BatchedGemmArgs& b, hipStream_t s);
int vz_launch_gemv(const LinearArgs& a, hipStream_t s);
int vz_launch_gemm128(const LinearArgs& a, hipStream_t s);
int vz_launch_gemm256(const LinearArgs& a, hipStream_t s);
int vz_init_gemm256_kernel();
int vz_gemm256_async_error(hipStream_t s, int* err, bool reset_only);
int vz_gemm256_corrupt_tickets(hipStream_t s, int tr, int arrive, int ready);
void vz_set_gemm_choice(int v);
void vz_set_splitk_mode(int v);
void vz_set_splitk_cap(int v);
void vz_set_splitk_mid(int v);   // 0 auto, 1 force 128x128, 2 force 256x256
int vz_launch_linear(const LinearArgs& a, hipStream_t s);  // picks by M

int vz_launch_layernorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const float* w, const float* b, int rows,
                        int cols, float eps, hipStream_t s);
int vz_launch_rmsnorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const float* w, int rows, int cols, float eps,
                      hipStream_t s);

struct AttnArgs {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int B, Sq, Sk, Hq, Hkv, head_dim;
    long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
    float scale;
    int causal, q_pos0, window;
    const int* kv_len;
    // optional fp32 workspace: lets a launch with few query rows and a long key range (Q-Former cross-attention: 32 x 576,
    // head_dim 512) split the keys over several workgroups and merge the partial softmaxes in a second kernel
    float* part = nullptr;
    size_t part_floats = 0;
    // optional (head_dim 128, the v2 kernel): q holds UNROTATED queries (e.g. the Q part of a fused QKV row) and the kernel applies RoPE
    // while it loads them - rope_pos[b * Sq + row] indexes the [pos][64] cos / sin tables; rope_kv_kernel's arithmetic, bit for bit
    const float *rope_cos = nullptr, *rope_sin = nullptr;
    const int* rope_pos = nullptr;
};
int vz_launch_attention(const AttnArgs& a, hipStream_t s);
void vz_set_attn_version(int v);
int vz_attn_version();
void vz_set_attn_split(int v);

// decode attention: one query token per slot against the KV cache; lengths live on the device
struct AttnDecodeArgs {
    const bf16_t* q;      // [B, Hq, D]
    const bf16_t* kc;     // cache base for this layer: [B][Hkv][max_ctx][D]
    const bf16_t* vc;
    bf16_t* o;            // [B, Hq, D]
    float* part;          // workspace [B*Hq*nsplit*(D+2)]
    int B, Hq, Hkv, D, max_ctx, nsplit, window;
    float scale;
    const int* ctx_len;   // device int32 [B]: keys visible to this step (incl. the token just appended)
};
int vz_launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s);

// decode_persist.hip: one resident grid per decoded token (batch 1, Zephyr-7B geometry, bf16 weights)
struct VzTokLayerHost { const void *qkv_w, *o_w, *gu_w, *down_w; const float *in_norm, *post_norm; void *kc, *vc; };
struct VzTokArgs {
    const void *embed, *lm_head; const float* final_norm;
    const int *cur, *pos, *slot, *step;
    float* logits; float* part; unsigned* ticket; const float *cosT, *sinT; int* err;
    int vocab, max_ctx, nsplit, window; float scale, eps;
};
struct VzTokState;
bool vz_decode_persist_supported();
int vz_decode_persist_create(const VzTokLayerHost* layers, int n_layers, VzTokState** out);
void vz_decode_persist_destroy(VzTokState* st);
int vz_decode_persist_reset(VzTokState* st, hipStream_t s);
int vz_decode_persist_poke(VzTokState* st, int word, unsigned value, hipStream_t s);
int vz_launch_decode_token(VzTokState* st, const VzTokArgs& a, hipStream_t s);
int vz_decode_persist_stamps(VzTokState* st, unsigned long long* host, int n_layers);

// comm_oneshot.hip: one-shot all-reduce of the tensor-parallel decode step (8-byte tagged granules into every peer's receive area)
size_t vz_oneshot_area_bytes(int n_ranks, int max_elems);
int vz_launch_allreduce_oneshot(void* const* areas, int rank, int n_ranks, int max_elems, const bf16_t* in, bf16_t* out, int n, unsigned* seq,
                                int* err, hipStream_t s);
int vz_launch_allreduce_oneshot_all(void* const* areas, int n_ranks, int max_elems, const bf16_t* const* in, bf16_t* const* out, int n,
                                    unsigned* const* seq, int* err, hipStream_t s);

// attn_bwd_flash.hip: tile-resident attention backward (head_dim 128; causal / window / padding masks; grouped KV heads).  dq bf16 strided
// like q; dk / dv [Sk, D] blocks per (batch, KV head), fp32 or bf16, same strides for both.  scratch: vz_flash_bwd_scratch_bytes.
struct FlashBwdArgs {
    const bf16_t *q, *k, *v, *dO;
    int B, Sq, Sk, Hq, Hkv, D;
    long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
    float scale; int causal, window; const int* kv_len;
    bf16_t* dq; long dq_bs, dq_ss, dq_hs;
    void *dk, *dv; int dkv_fp32; long dk_bs, dk_ss, dk_hs;
};
size_t vz_flash_bwd_scratch_bytes(int B, int Sq, int Hq);
bool vz_flash_bwd_ok(const FlashBwdArgs& a);
int vz_launch_flash_bwd(const FlashBwdArgs& a, void* scratch, size_t scratch_bytes, hipStream_t s);

// decode attention with RoPE + KV append + split combine fused into one launch (attn_decode.hip)
struct AttnDecodeFusedArgs {
    const bf16_t* qkv;     // [B, (Hq+2Hkv)*D]
    bf16_t *kc, *vc, *o;   // caches for this layer [B][Hkv][max_ctx][D]; o [B,Hq,D]
    float* part;           // [B*Hkv*nsplit*(4*D+32)]
    unsigned* ticket;      // [B*Hkv], zeroed once
    const float *cosT, *sinT;
    const int *pos, *slot; // device int32 [B]
    int B, Hq, Hkv, D, max_ctx, nsplit, window;
    float scale;
};
int vz_launch_attn_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s);
// attn_o_fused.hip: the same attention + the O projection (x += att . o_w^T, batch 1) in one launch; `done` = a zeroed device word,
// `step` = the decode call's device-side step counter
int vz_launch_attn_o_fused(const AttnDecodeFusedArgs& a, const bf16_t* o_w, const unsigned char* o_w8, const float* o_scale, bf16_t* att_scratch,
                           bf16_t* x, unsigned* done, const int* step, int layer, int n_layers, int* err, hipStream_t s);


int vz_launch_rope_kv(const bf16_t* qkv, int ld, bf16_t* q_out, bf16_t* kc, bf16_t* vc, const float* cosT,
                      const float* sinT, const int* pos, const int* slot, int B, int S, int Hq, int Hkv, int D,
                      int max_ctx, hipStream_t s);
int vz_launch_gather_rows(const int* kind, const int* idx, int rows, int cols, const bf16_t* table,
                          const bf16_t* visual, bf16_t* out, hipStream_t s);
int vz_launch_embed_tokens(const int* ids, int rows, int cols, const bf16_t* table, bf16_t* out, hipStream_t s);
int vz_launch_im2col(const bf16_t* img, int T, int image, int patch, int kpad, bf16_t* out, hipStream_t s);
int vz_launch_clip_assemble(const bf16_t* patch_out, const bf16_t* cls, const bf16_t* pos, int T, int tokens, int C,
                            bf16_t* out, hipStream_t s);
int vz_launch_fusion(const bf16_t* hs_base, long layer_stride, int first_layer, int groups, int per_group, int T,
                     int tokens, int C, int skip, bf16_t* out, hipStream_t s);
int vz_launch_argmax(const float* logits, int rows, int cols, int* ids, int* pos, int* slot, int* len, int* out_ids,
                     int out_stride, const int* step, int max_ctx, int rope_max, int* ring, int ring_n, hipStream_t s);
int vz_launch_sample(const float* logits, int rows, int cols, float temperature, int top_k, float top_p, const unsigned* seed,
                     const int* ctr, int ctr_add, int* ids, int* pos, int* slot, int* len, int* out_ids, int out_stride,
                     const int* step, int max_ctx, int rope_max, int* ring, int ring_n, hipStream_t s);
// set dynamic-LDS limits of every kernel up front (never inside a stream capture)
int vz_init_gemm_kernels();
int vz_init_attention_kernels();
int vz_init_sampling_kernels();
int vz_launch_copy_rows(const bf16_t* src, long src_stride, bf16_t* dst, long dst_stride, int rows, int cols,
                        hipStream_t s);
// KV-cache row moves (batched admissions of the continuous-batching loop): up to 16 (src row, dst row, tokens) triples per launch
struct KvMoves { int n; int src[16], dst[16], len[16]; };
int vz_launch_kv_move_rows(bf16_t* kv, size_t layer_elems, int n_layers, int max_batch, int Hkv, int max_ctx, int D, const KvMoves& mv,
                           hipStream_t s);
int vz_launch_step_advance(int* step, hipStream_t s);
int vz_launch_repack_logits(const float* gathered, float* out, int rows, int Vp, int V, int tp, hipStream_t s);

// ---- backward kernels of the Stage-1 training step (train.hip) ----
int vz_launch_transpose(const bf16_t* src, long src_rs, long src_so, long src_si, bf16_t* dst, long dst_rs, long dst_so, long dst_si, int R,
                        int C, int n_outer, int n_inner, int col0, hipStream_t s);
int vz_launch_softmax_fwd(const float* S, int lds_, bf16_t* P, int ldp, long rows, int H, int Sq, int Sk, float scale, int causal, int window,
                          const int* kv_len, hipStream_t s);
int vz_launch_softmax_bwd(const bf16_t* P, int ldp, const float* dP, int lddp, bf16_t* dS, int ldds, long rows, int Sk, float scale, hipStream_t s);
int vz_launch_rmsnorm_bwd(const bf16_t* x, const float* w, const bf16_t* dy, const bf16_t* dres, bf16_t* dx, long rows, int cols, float eps,
                          hipStream_t s);
int vz_layernorm_bwd_groups(long rows);
size_t vz_layernorm_bwd_scratch_floats(long rows, int cols);
int vz_launch_layernorm_bwd(const bf16_t* x, const float* w, const bf16_t* dy, const bf16_t* dres, bf16_t* dx, float* part, float* dw, float* db,
                            long rows, int cols, float eps, hipStream_t s);
int vz_launch_gelu_fwd(const bf16_t* h, bf16_t* y, long n, hipStream_t s);
int vz_launch_gelu_bwd(const bf16_t* h, const bf16_t* dy, bf16_t* dh, long n, hipStream_t s);
int vz_launch_swiglu_fwd(const bf16_t* gu, bf16_t* act, long rows, int I, hipStream_t s);
int vz_launch_swiglu_bwd(const bf16_t* gu, const bf16_t* dact, bf16_t* dgu, long rows, int I, hipStream_t s);
int vz_launch_rope_bwd_assemble(const bf16_t* dq, const float* dk, const float* dv, bf16_t* dqkv, const float* cosT, const float* sinT, const int* pos,
                                int B, int S, int Hq, int Hkv, int D, int Sk_ld, hipStream_t s);
int vz_launch_causal_lm_loss(const float* logits, int B, int S, int V, const int* labels, float* loss_rows, float* out, hipStream_t s);
int vz_launch_cross_entropy(const float* logits, int V, const int* labels, long rows, int S, float inv_n, float* loss_rows, bf16_t* dlogits, int ldd,
                            hipStream_t s);
int vz_colsum_groups(long rows);
int vz_launch_colsum(const bf16_t* y, int ld, long rows, int cols, float* part, float* out, hipStream_t s);
int vz_launch_gather_rows_idx(const bf16_t* src, const int* idx, bf16_t* dst, long rows, int cols, hipStream_t s);
int vz_launch_segment_sum_rows(const bf16_t* src, const int* map, int n_src, int rows_per, bf16_t* dst, int n_dst, int cols, hipStream_t s);
int vz_launch_axpy_f32(float* y, const float* x, long n, hipStream_t s);
int vz_launch_add_bf16(bf16_t* y, const bf16_t* x, long n, hipStream_t s);
int vz_launch_f32_to_bf16(const float* x, bf16_t* y, long n, hipStream_t s);
int vz_launch_bf16_to_f32(const bf16_t* x, float* y, long n, hipStream_t s);
int vz_launch_acc_rows_f32(float* out, const bf16_t* src, int n_batches, long stride, int rows, int cols, hipStream_t s);
int vz_launch_adamw(float* p, float* m, float* v, float* g, void* work, int work_bf16, long n, float lr, float b1, float b2, float eps, float wd,
                    int t, hipStream_t s);
