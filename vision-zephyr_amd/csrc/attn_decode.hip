// Decode-step attention for Zephyr-7B (gfx950), one launch per layer:
//   RoPE on the new token's Q and K (rotate-half, hf:models/mistral/modeling_mistral.py:51-81)
//   + append of K/V to the cache + softmax(q K^T) V over the cache (hf:...:139-178, GQA, sliding window)
//   + the combine of the context splits, all in ONE kernel.
//
// HBM-bound: per layer and token the kernel streams the KV cache once (2 * 8 * ctx * 256 B = 8.4 MB at
// ctx 2048).  Grid = (nsplit, Hkv, B): a workgroup owns one KV head and one slice of the context and
// serves the 4 query heads of that KV head from a single pass over K and V (GQA without repeat_kv, K/V
// read once instead of 4x).  K/V rows are read with 16-byte loads straight to VGPRs, 16 lanes per
// 256-byte row, 16 rows in flight per workgroup pass; the tiny per-split partials (m, l, o[4][128]) go
// through HBM and the LAST workgroup to arrive for a (slot, kv head) merges them
// (agent-scope release -> ticket -> acquire; placement independent, cdna guide section 6 G16).
// Every per-step quantity (position, cache slot, visible length) is read from device memory, so one
// captured hipGraph replays for every token.
#include "vz_common.h"

namespace {

constexpr int D = 128;          // head_dim
constexpr int G = 4;            // query heads per KV head (32 / 8)
constexpr int CH = 128;         // keys per inner chunk
constexpr int PW = G * (D + 2); // floats per partial record: per head [m, l, o[128]]

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
    const int per = (span + p.nsplit - 1) / p.nsplit;
    const int k0 = lo + split * per, k1 = min(len, k0 + per);
    const int heads = p.Hq + 2 * p.Hkv;
    const bf16_t* row = p.qkv + (size_t)b * heads * D;

    // ---- RoPE: thread (h = tid>>6, pair = lane) rotates (d, d+64) of query head hk*G + h ----
    {
        const float c = p.cosT[(size_t)position * (D / 2) + lane], s = p.sinT[(size_t)position * (D / 2) + lane];
        const bf16_t* qh = row + (size_t)(hk * G + wave) * D;
        const float x = bf16_to_f32(qh[lane]), y = bf16_to_f32(qh[lane + 64]);
        // round to bf16 exactly like the separate RoPE kernel does before the attention consumes it
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
    bf16_t* kb = p.kc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;
    bf16_t* vb = p.vc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;
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
        // ---- every K and V row of the chunk goes in flight first (8 + 8 x 16 B per lane): the scores and the
        //      softmax run under the V loads' latency ----
        uint4 kreg[CH / 16], vreg[CH / 16];
#pragma unroll
        for (int i = 0; i < CH / 16; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (kk < n) kreg[i] = kidx == slot ? *(const uint4*)(knew + sub * 8) : *(const uint4*)(kb + (size_t)kidx * D + sub * 8);
        }
#pragma unroll
        for (int i = 0; i < CH / 16; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (kk < n) vreg[i] = kidx == slot ? *(const uint4*)(vnew + sub * 8) : *(const uint4*)(vb + (size_t)kidx * D + sub * 8);
        }
        // ---- scores ----
#pragma unroll
        for (int i = 0; i < CH / 16; ++i) {
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
        for (int i = 0; i < CH / 16; ++i) {
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
    // ---- reduce the 16 key slots through LDS, write this split's partial ----
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
        po[h * (D + 2) + 2 + d] = v;
    }
    if (tid < G) { po[tid * (D + 2)] = stat[G + tid]; po[tid * (D + 2) + 1] = stat[2 * G + tid]; }

    // ---- publish + ticket; the last arriver for (b, hk) merges all splits ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(p.ticket + (size_t)b * p.Hkv + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = (t == (unsigned)p.nsplit - 1) ? 1u : 0u;
        if (last_flag) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_flag) return;
    // ---- merge: weights w[h][s] = exp(m_s - m) in LDS first, then every thread sums its (h, d) over the splits
    //      with the loads of all splits independent (8 in flight) ----
    const float* pp = p.part + ((size_t)b * p.Hkv + hk) * p.nsplit * PW;
    float* wgt = &red[0][0][0];            // [G][64] weights, then [G] 1/l   (red is free again)
    if (tid < G * 64) {
        const int h = tid >> 6, s2 = tid & 63;
        float ms = -INFINITY, ls = 0.f;
        if (s2 < p.nsplit) { ms = pp[(size_t)s2 * PW + h * (D + 2)]; ls = pp[(size_t)s2 * PW + h * (D + 2) + 1]; }
        const float m = wave_max(ms);
        const float w = ms == -INFINITY ? 0.f : __expf(ms - m);
        const float l = wave_sum(w * ls);
        wgt[h * 64 + s2] = w;
        if (s2 == 0) wgt[G * 64 + h] = l > 0.f ? 1.0f / l : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < G * D; i += 256) {
        const int h = i >> 7, d = i & 127;
        float a = 0.f;
#pragma unroll 8
        for (int s2 = 0; s2 < p.nsplit; ++s2) a += wgt[h * 64 + s2] * pp[(size_t)s2 * PW + h * (D + 2) + 2 + d];
        p.o[((size_t)b * p.Hq + hk * G + h) * D + d] = f32_to_bf16(a * wgt[G * 64 + h]);
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
