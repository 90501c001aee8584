// Decode attention + O projection of a 1- or 2-row step in ONE launch (gfx950).
//
// In the launch chain the O projection's GEMV (33.5 MB of weights, ~8 us) starts after the attention kernel (~12.8 us at ctx 2048, a chain
// of dependent round trips during which HBM is nearly idle) has ended: 20.8 us for 42 MB.  A weight never depends on the token, so here
// the O projection's workgroups are part of the attention's grid: they request their two weight rows per wave at once (the whole matrix:
// 2048 waves x 16 KiB) and only then wait for the attention's eight mergers - the stream runs UNDER the attention's latency chain.
//   * blocks [0, nsplit * Hkv): the stand-alone attention kernel's body (attn_decode_body.h, same arithmetic, same bits); a merger stores
//     its four heads write-through, drains, and adds one to the arrival word;
//   * blocks [nsplit * Hkv, + 512): four waves each, two rows of the O projection per wave: weights to registers (non-temporal), one wave
//     polls the arrival word (agent scope, bounded), the block gathers the 4096 attention outputs with L2-bypassing loads into LDS and
//     runs the GEMV's dot products + residual epilogue (gemv_bf16_kernel's lane -> k assignment and chunk order: same bits).
// 256-thread blocks at <= 168 registers: three per CU, 768 slots for at most 256 + 512 blocks - every block of the grid is resident at
// once, so a waiting block can only wait for blocks that are running (no order or placement is assumed; the wait is bounded anyway:
// an expired one raises the async error word).  The arrival word is monotonic within a vz_llm_decode_steps call
// (target = (step * layers + layer + 1) * Hkv) and zeroed by the host with the step counter.
// Two rows (NB = 2): the rows' attention workgroups side by side (rows x splits <= 32), both rows' outputs gathered, 2 x 2 dot products per wave:
// 3.11 -> 2.99 ms per 2-row step.
// Measured (bench.py's request, ctx 2048): 357 tok/s against 340 for the two launches (attention 12.6 us + O GEMV 8.5 us -> 19.4 us for
// the fused launch).  Also built and measured this round, and removed again: the RMSNorm + QKV projection as a THIRD role in front
// (QKV role one block per CU, the cached K / V requested before QKV had finished, the O role waiting for QKV before it streams): bit-identical,
// 30.3 us against 11.3 + 19.4 - the QKV role's slowest block finished at ~16 us (a CU's memory pipe accepts only ~25-60 GB/s and the role
// shares it with the other roles' K / V loads and polls), which ate what the earlier K / V loads and the saved boundary gave.
// Hand-off form: cdna guide section 6 G16 / MI355X notes, measured row 1 (sc1 payload stores drained by every storing wave, workgroup
// barrier, one agent-scope add; consumer: sc1 poll, workgroup barrier, sc1 loads).
#include "attn_decode_body.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
using namespace attn_dec;

constexpr int OH = 4096;                 // hidden = rows and K of the O projection
constexpr int O_BLOCKS = 512;            // x 4 waves x 2 rows
constexpr int SPIN_CAP = 400000;

struct AoParams {
    FusedParams at;                      // attention (o = the hand-off vector [4096] bf16)
    const bf16_t* o_w;                   // [4096][4096]
    const unsigned char* o_w8;           // FP8 instantiation: e4m3 rows [4096][4096 bytes] + one 2^e scale per row (gemv.hip's W8A16 stream)
    const float* o_scale;
    bf16_t* x;                           // residual stream [4096]: x += att . o_w^T (in place)
    unsigned* done;                      // arrival word of the mergers
    const int* step;                     // device-side step counter of the decode call
    int* err;
    int layer, n_layers;
    int delay;                           // O role: units of ~0.21 us (s_sleep 8) to wait before requesting the weights (experiment knob 31)
};

__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w0), __builtin_bit_cast(bf16x2, x0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w1), __builtin_bit_cast(bf16x2, x1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w2), __builtin_bit_cast(bf16x2, x2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w3), __builtin_bit_cast(bf16x2, x3), acc, false);
    return acc;
}

// 16 e4m3 weights of one lane -> 16 bf16 (k order preserved; exact) - gemv.hip's conversion
__device__ __forceinline__ void fp8x16_to_bf16(const u32x4 w, u32x4& lo, u32x4& hi) {
    auto cv = [](unsigned v, bool hi_half) {
        return hi_half ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, true))
                       : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, false));
    };
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lo = (u32x4){cv(w0, false), cv(w0, true), cv(w1, false), cv(w1, true)};
    hi = (u32x4){cv(w2, false), cv(w2, true), cv(w3, false), cv(w3, true)};
}

union AoShared {
    Shared attn;
    struct { __attribute__((aligned(16))) bf16_t att[2 * OH]; int ok; } o;      // up to two rows' attention outputs
};

template <bool FP8, int NB>          // NB = rows of the decode step (1 or 2: the GEMV route's row counts that keep the whole grid resident)
__global__ __launch_bounds__(256, 3) void attn_o_fused_kernel(AoParams p) {      // 3 waves per SIMD = 3 blocks per CU: <= 168 registers
    __shared__ AoShared sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per_row = p.at.nsplit * p.at.Hkv, n_attn = NB * per_row;
    if ((int)blockIdx.x < n_attn) {
        // ---- attention role ----
        const int b = NB == 1 ? 0 : (int)blockIdx.x / per_row, rem = (int)blockIdx.x - b * per_row;
        const int split = rem % p.at.nsplit, hk = rem / p.at.nsplit;
        const bool merged = body<true>(p.at, p.at.qkv + (size_t)b * (p.at.Hq + 2 * p.at.Hkv) * D, split, hk, b, tid, true, sm.attn);
        if (!merged) return;                                   // (uniform per block)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the merged heads' write-through stores have left
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- O-projection role: rows r0, r0 + 1 of this wave ----
    const int ob = blockIdx.x - n_attn;
    const int r0 = (ob * 4 + wave) * 2;
    for (int i = 0; i < p.delay; ++i) __builtin_amdgcn_s_sleep(8);
    constexpr int NC = FP8 ? 4 : 8;                          // 16-byte pieces per lane and row: 8 x 8 bf16 or 4 x 16 e4m3
    u32x4 w[2][NC];
    if constexpr (FP8) {
        const unsigned char* wp = p.o_w8 + (size_t)r0 * OH + lane * 16;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) w[r][c] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)r * OH + c * 1024));
    } else {
        const bf16_t* wp = p.o_w + (size_t)r0 * OH + lane * 8;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) w[r][c] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)r * OH + c * 512));
    }
    unsigned resid[NB];                                        // x[b][r0], x[b][r0 + 1]
#pragma unroll
    for (int b = 0; b < NB; ++b) resid[b] = *(const unsigned*)(p.x + (size_t)b * OH + r0);
    const unsigned target = ((unsigned)p.step[0] * (unsigned)p.n_layers + (unsigned)p.layer + 1u) * (unsigned)(p.at.Hkv * NB);
    if (wave == 0) {
        int ok = 0;
        for (int it = 0; it < SPIN_CAP; ++it) {
            const unsigned v = __hip_atomic_load(p.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(v - target) >= 0) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (lane == 0) { sm.o.ok = ok; if (!ok) atomicExch(p.err, VZ_ASYNC_ATTN_O); }
    }
    __syncthreads();
    // every merged head is in memory: gather the 4096 outputs (8-byte L2-bypassing loads, all in flight), then the dot products
    {
        unsigned long long v[4 * NB];        // (the rows' outputs are contiguous in the hand-off vector: [NB][4096])
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) v[i] = __hip_atomic_load((const unsigned long long*)(p.at.o + (i * 256 + tid) * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4 * NB; ++i) *(unsigned long long*)(sm.o.att + (i * 256 + tid) * 4) = v[i];
    }
    __syncthreads();
    float acc[2][NB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
    if constexpr (FP8) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {                       // gemv_bf16_kernel<.., FP8>'s chunk order and lane -> k assignment (1024 k per chunk, 16 per lane)
            u32x4 wl[2], wh[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) fp8x16_to_bf16(w[r][c], wl[r], wh[r]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bf16_t* xp = sm.o.att + b * OH + c * 1024 + lane * 16;
                const u32x4 x0 = *(const u32x4*)xp, x1 = *(const u32x4*)(xp + 8);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][b] = dot8(wh[r], x1, dot8(wl[r], x0, acc[r][b]));
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const u32x4 xv = *(const u32x4*)(sm.o.att + b * OH + c * 512 + lane * 8);
#pragma unroll
                for (int r = 0; r < 2; ++r) acc[r][b] = dot8(w[r][c], xv, acc[r][b]);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float s0 = wave_sum_lane63(acc[0][b]), s1 = wave_sum_lane63(acc[1][b]);
        if (lane == 63) {
            float t0 = s0, t1 = s1;
            if constexpr (FP8) { t0 *= p.o_scale[r0]; t1 *= p.o_scale[r0 + 1]; }       // the row's power-of-two scale, once per output
            t0 += bf16_to_f32((unsigned short)(resid[b] & 0xffffu)); t1 += bf16_to_f32((unsigned short)(resid[b] >> 16));
            if (!sm.o.ok) t0 = t1 = __uint_as_float(0x7fc00000u);          // an expired wait never passes for a result
            *(unsigned*)(p.x + (size_t)b * OH + r0) = pack_bf16x2(t0, t1);
        }
    }
}

}  // namespace

int g_attn_o_delay = 12;      // vz_tune_set(31, n): the O role waits n x ~0.21 us before requesting its weights, so that the attention blocks' K / V loads reach the memory system first (scan at ctx 2048: 0 -> 344, 8 -> 346, 12 -> 357, 16 -> 352, 24 -> 348 tok/s)
int vz_launch_attn_o_fused(const AttnDecodeFusedArgs& a, const bf16_t* o_w, const unsigned char* o_w8, const float* o_scale, bf16_t* att_scratch,
                           bf16_t* x, unsigned* done, const int* step, int layer, int n_layers, int* err, hipStream_t s) {
    VZ_CHECK_ARG((a.B == 1 || a.B == 2) && a.D == D && a.Hq == 32 && a.Hkv == 8 && a.nsplit >= 1 && a.B * a.nsplit <= 32,
                 "attn_o_fused: 1 or 2 rows, 32 / 8 heads of 128, rows x context splits <= 32 (every workgroup resident)");
    VZ_CHECK_ARG(a.qkv && a.kc && a.vc && att_scratch && a.part && a.ticket && (o_w || (o_w8 && o_scale)) && x && done && step && err, "attn_o_fused: null argument");
    AoParams p;
    p.at.qkv = a.qkv; p.at.kc = a.kc; p.at.vc = a.vc; p.at.o = att_scratch; p.at.part = a.part; p.at.ticket = a.ticket;
    p.at.cosT = a.cosT; p.at.sinT = a.sinT; p.at.pos = a.pos; p.at.slot = a.slot;
    p.at.B = a.B; p.at.Hq = a.Hq; p.at.Hkv = a.Hkv; p.at.max_ctx = a.max_ctx; p.at.nsplit = a.nsplit; p.at.window = a.window; p.at.scale = a.scale;
    p.o_w = o_w; p.o_w8 = o_w8; p.o_scale = o_scale; p.x = x; p.done = done; p.step = step; p.err = err; p.layer = layer; p.n_layers = n_layers; p.delay = g_attn_o_delay;
    const dim3 grid(a.B * a.nsplit * a.Hkv + O_BLOCKS);
    if (o_w8 && o_scale) {       // e4m3 rows take precedence (as in linear())
        if (a.B == 1) hipLaunchKernelGGL((attn_o_fused_kernel<true, 1>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_o_fused_kernel<true, 2>), grid, dim3(256), 0, s, p);
    } else {
        if (a.B == 1) hipLaunchKernelGGL((attn_o_fused_kernel<false, 1>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((attn_o_fused_kernel<false, 2>), grid, dim3(256), 0, s, p);
    }
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

