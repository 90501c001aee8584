// Decode-step attention for Zephyr-7B (gfx950), one launch per layer:
//   RoPE on the new token's Q and K (rotate-half, hf:models/mistral/modeling_mistral.py:51-81)
//   + append of K/V to the cache + softmax(q K^T) V over the cache (hf:...:139-178, GQA, sliding window)
//   + the combine of the context splits, all in ONE kernel.
//
// HBM/latency-bound: per layer and token the kernel streams the KV cache once (2 * 8 * ctx * 256 B = 8.4 MB
// at ctx 2048) and sits on the critical path of every decoded token, so the design minimises the dependent
// chain rather than the byte count:
//   * grid (nsplit, Hkv, B): a workgroup owns one KV head and one slice of the context and serves the 4 query
//     heads of that KV head from a single pass over K and V (GQA without repeat_kv, K/V read once, not 4x);
//   * the K and V rows of the slice are requested FIRST (16-byte loads straight to VGPRs, 16 lanes per
//     256-byte row, up to 8+8 rows per lane in flight); the RoPE of the new token runs under that latency;
//   * the per-split partials go to HBM with write-through (sc1) stores, one relaxed agent-scope ticket per
//     workgroup, and the LAST workgroup to arrive for a (slot, kv head) merges them with sc1 loads - no
//     release/acquire cache maintenance on the critical path (CDNA4 guide section 6 G16, valid form "sc1 payload +
//     drained ticket, last arriver told by the value its add returned"); placement independent.
// Every per-step quantity (position, cache slot) is read from device memory, so one captured hipGraph
// replays for every token.
#include "vz_common.h"

namespace {

constexpr int D = 128;            // head_dim
constexpr int G = 4;              // query heads per KV head (32 / 8)
constexpr int CH = 128;           // keys per inner chunk
constexpr int NR = CH / 16;       // K (and V) rows per lane per chunk
constexpr int PW = G * D + 32;    // floats per partial record: o[4][128] | m[4] l[4] pad  (17 x 128-byte lines)

typedef unsigned __attribute__((address_space(1))) gu32;

__device__ __forceinline__ void st_sc1(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
    return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

struct FusedParams {
    const bf16_t* qkv;    // [B, (Hq + 2 Hkv) * D] fresh projection of the new token
    bf16_t* kc;           // [B][Hkv][max_ctx][D]
    bf16_t* vc;
    bf16_t* o;            // [B, Hq, D]
    float* part;          // [B][Hkv][nsplit][PW]
    unsigned* ticket;     // [B][Hkv], zero before the first launch; the last arriver re-zeroes it
    const float* cosT;    // [max_pos, D/2]
    const float* sinT;
    const int* pos;       // [B] position id of the new token
    const int* slot;      // [B] cache slot it is written to (= tokens already cached)
    int B, Hq, Hkv, max_ctx, nsplit, window;
    float scale;
};

__global__ __launch_bounds__(256) void attn_decode_fused_kernel(FusedParams p) {
    __shared__ __attribute__((aligned(16))) float q_s[G][D];          // rotated, pre-scaled queries
    __shared__ __attribute__((aligned(16))) bf16_t knew[D], vnew[D];  // the new token's (rotated) K and V
    __shared__ __attribute__((aligned(16))) float sc[G][CH];          // scores -> probabilities of the chunk
    __shared__ float stat[3 * G];                                     // per head: alpha | m_run | l_run
    __shared__ __attribute__((aligned(16))) float red[16][G][D];      // PV partial sums per key slot (32 KiB)
    __shared__ unsigned last_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int sub = lane & 15, ks = wave * 4 + (lane >> 4);           // 8-wide d chunk, key slot 0..15
    const int slot = p.slot[b], len = slot + 1, position = p.pos[b];
    const int lo = p.window > 0 ? max(0, len - p.window) : 0;
    const int span = len - lo;
    // a split takes at least one chunk: a short context (a 16-row batch at ctx 100) is served by ONE workgroup per KV head
    // and row instead of nsplit mostly empty ones all running the ticket protocol; the others leave at once
    const int per = max(CH, (span + p.nsplit - 1) / p.nsplit);
    const int n_active = (span + per - 1) / per;
    if (split >= n_active) return;
    const int k0 = lo + split * per, k1 = min(len, k0 + per);
    const int heads = p.Hq + 2 * p.Hkv;
    const bf16_t* row = p.qkv + (size_t)b * heads * D;
    bf16_t* kb = p.kc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;
    bf16_t* vb = p.vc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;

    // ---- the first chunk's K and V rows go in flight before anything else (the new token's row, index `slot`,
    //      is not in the cache yet: it is patched in from LDS after the RoPE) ----
    uint4 kreg[NR], vreg[NR];
    auto issue = [&](int c0, int n) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (kk < n && kidx != slot) kreg[i] = *(const uint4*)(kb + (size_t)kidx * D + sub * 8);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (kk < n && kidx != slot) vreg[i] = *(const uint4*)(vb + (size_t)kidx * D + sub * 8);
        }
    };
    if (k0 < k1) issue(k0, min(CH, k1 - k0));

    // ---- RoPE: thread (h = wave, pair = lane) rotates (d, d+64) of query head hk*G + h ----
    {
        const float c = p.cosT[(size_t)position * (D / 2) + lane], s = p.sinT[(size_t)position * (D / 2) + lane];
        const bf16_t* qh = row + (size_t)(hk * G + wave) * D;
        const float x = bf16_to_f32(qh[lane]), y = bf16_to_f32(qh[lane + 64]);
        // rounded to bf16 exactly like the stand-alone RoPE kernel before the attention consumes it
        q_s[wave][lane] = bf16_to_f32(f32_to_bf16(x * c - y * s)) * p.scale;
        q_s[wave][lane + 64] = bf16_to_f32(f32_to_bf16(y * c + x * s)) * p.scale;
        if (wave == 0) {
            const bf16_t* kh = row + (size_t)(p.Hq + hk) * D;
            const float kx = bf16_to_f32(kh[lane]), ky = bf16_to_f32(kh[lane + 64]);
            knew[lane] = f32_to_bf16(kx * c - ky * s);
            knew[lane + 64] = f32_to_bf16(ky * c + kx * s);
        } else if (wave == 1) {
            const bf16_t* vh = row + (size_t)(p.Hq + p.Hkv + hk) * D;
            vnew[lane] = vh[lane];
            vnew[lane + 64] = vh[lane + 64];
        }
    }
    if (tid < G) { stat[G + tid] = -INFINITY; stat[2 * G + tid] = 0.f; }
    __syncthreads();
    if (split == 0 && tid < 32) {   // one workgroup per (slot, kv head) appends the new row to the cache
        if (tid < 16) *(uint4*)(kb + (size_t)slot * D + tid * 8) = *(const uint4*)(knew + tid * 8);
        else *(uint4*)(vb + (size_t)slot * D + (tid - 16) * 8) = *(const uint4*)(vnew + (tid - 16) * 8);
    }

    float qr[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) qr[h][j] = q_s[h][sub * 8 + j];
    float acc[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;

    for (int c0 = k0; c0 < k1; c0 += CH) {
        const int n = min(CH, k1 - c0);
        if (c0 != k0) issue(c0, n);
        if (slot >= c0 && slot < c0 + n && ((slot - c0) & 15) == ks) {   // this lane group holds the new token's row
            const int i_new = (slot - c0) >> 4;
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (i == i_new) { kreg[i] = *(const uint4*)(knew + sub * 8); vreg[i] = *(const uint4*)(vnew + sub * 8); }
        }
        // ---- scores ----
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i;
            const u16x8 kv = __builtin_bit_cast(u16x8, kreg[i]);
            float s[G] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float kf = bf16_to_f32(kv[j]);
#pragma unroll
                for (int h = 0; h < G; ++h) s[h] += qr[h][j] * kf;
            }
#pragma unroll
            for (int h = 0; h < G; ++h) {
                s[h] += __shfl_xor(s[h], 1, 64); s[h] += __shfl_xor(s[h], 2, 64);
                s[h] += __shfl_xor(s[h], 4, 64); s[h] += __shfl_xor(s[h], 8, 64);
            }
            if (sub == 0 && kk < n) {
#pragma unroll
                for (int h = 0; h < G; ++h) sc[h][kk] = s[h];
            }
        }
        __syncthreads();
        // ---- online softmax, wave h owns head h ----
        {
            const float s0 = lane < n ? sc[wave][lane] : -INFINITY;
            const float s1 = lane + 64 < n ? sc[wave][lane + 64] : -INFINITY;
            const float m_old = stat[G + wave];
            const float m_new = fmaxf(m_old, wave_max(fmaxf(s0, s1)));
            const float e0 = __expf(s0 - m_new), e1 = __expf(s1 - m_new);
            if (lane < n) sc[wave][lane] = e0;
            if (lane + 64 < n) sc[wave][lane + 64] = e1;
            const float ps = wave_sum(e0 + e1);
            const float alpha = __expf(m_old - m_new);
            if (lane == 0) { stat[wave] = alpha; stat[G + wave] = m_new; stat[2 * G + wave] = stat[2 * G + wave] * alpha + ps; }
        }
        __syncthreads();
        // ---- O += P V ----
        float al[G];
#pragma unroll
        for (int h = 0; h < G; ++h) al[h] = stat[h];
#pragma unroll
        for (int h = 0; h < G; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[h][j] *= al[h];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i;
            if (kk < n) {
                const u16x8 vv = __builtin_bit_cast(u16x8, vreg[i]);
                float pr[G];
#pragma unroll
                for (int h = 0; h < G; ++h) pr[h] = sc[h][kk];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float vf = bf16_to_f32(vv[j]);
#pragma unroll
                    for (int h = 0; h < G; ++h) acc[h][j] += pr[h] * vf;
                }
            }
        }
        __syncthreads();   // sc / stat are rewritten by the next chunk
    }
    // ---- reduce the 16 key slots through LDS, write this split's partial (write-through stores) ----
#pragma unroll
    for (int h = 0; h < G; ++h) {
        *(f32x4*)&red[ks][h][sub * 8] = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
        *(f32x4*)&red[ks][h][sub * 8 + 4] = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
    }
    __syncthreads();
    float* po = p.part + (((size_t)b * p.Hkv + hk) * p.nsplit + split) * PW;
    for (int i = tid; i < G * D; i += 256) {
        const int h = i >> 7, d = i & 127;
        float v = 0.f;
#pragma unroll
        for (int s16 = 0; s16 < 16; ++s16) v += red[s16][h][d];
        st_sc1(po + i, v);
    }
    if (tid < G) { st_sc1(po + G * D + tid, stat[G + tid]); st_sc1(po + G * D + G + tid, stat[2 * G + tid]); }

    // ---- every storing wave drains its stores, then ONE lane takes the ticket; the last arriver merges ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.ticket + (size_t)b * p.Hkv + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (t == (unsigned)n_active - 1) ? 1u : 0u;
    }
    __syncthreads();   // the wave that added joins this barrier after its add returned; everyone loads behind it
    if (!last_flag) return;
    const float* pp = p.part + ((size_t)b * p.Hkv + hk) * p.nsplit * PW;
    float* wgt = &red[0][0][0];            // [G][64] weights, then [G] 1/l   (red is free again)
    {
        const int h = tid >> 6, s2 = tid & 63;
        float ms = -INFINITY, ls = 0.f;
        if (s2 < n_active) { ms = ld_sc1(pp + (size_t)s2 * PW + G * D + h); ls = ld_sc1(pp + (size_t)s2 * PW + G * D + G + h); }
        const float m = wave_max(ms);
        const float w = ms == -INFINITY ? 0.f : __expf(ms - m);
        const float l = wave_sum(w * ls);
        wgt[h * 64 + s2] = w;
        if (s2 == 0) wgt[G * 64 + h] = l > 0.f ? 1.0f / l : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < G * D; i += 256) {
        const int h = i >> 7;
        float a = 0.f;
#pragma unroll 8
        for (int s2 = 0; s2 < n_active; ++s2) a += wgt[h * 64 + s2] * ld_sc1(pp + (size_t)s2 * PW + i);
        p.o[((size_t)b * p.Hq + hk * G) * D + i] = f32_to_bf16(a * wgt[G * 64 + h]);
    }
    if (tid == 0) __hip_atomic_store(p.ticket + (size_t)b * p.Hkv + hk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

int vz_launch_attn_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(a.D == D && a.Hq == a.Hkv * G, "attn_decode_fused: needs head_dim 128 and 4 query heads per KV head");
    VZ_CHECK_ARG(a.nsplit >= 1 && a.nsplit <= 64 && a.qkv && a.kc && a.vc && a.o && a.part && a.ticket, "attn_decode_fused: bad argument");
    FusedParams p;
    p.qkv = a.qkv; p.kc = a.kc; p.vc = a.vc; p.o = a.o; p.part = a.part; p.ticket = a.ticket;
    p.cosT = a.cosT; p.sinT = a.sinT; p.pos = a.pos; p.slot = a.slot;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.max_ctx = a.max_ctx; p.nsplit = a.nsplit; p.window = a.window; p.scale = a.scale;
    hipLaunchKernelGGL(attn_decode_fused_kernel, dim3(a.nsplit, a.Hkv, a.B), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
