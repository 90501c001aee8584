// Body of the decode-step attention (RoPE of the new token + KV append + softmax(q K^T) V over the cache + the merge of the context
// splits), shared by the stand-alone kernel (attn_decode.hip: one launch per layer) and the persistent decode-token kernel
// (decode_persist.hip: a phase of the resident grid).  One workgroup = (context split, KV head, batch row); 256 ACTIVE threads.
// In the persistent kernel the workgroup has more waves than that: the extra waves pass `act = false` - they take part in every
// workgroup barrier (the barrier count is uniform: it depends on the split's key range and on last_flag only) and touch no memory.
// Arithmetic and summation orders are the stand-alone kernel's, so both paths give the same bits.
#pragma once
#include "vz_common.h"

namespace attn_dec {

constexpr int D = 128;            // head_dim
constexpr int G = 4;              // query heads per KV head (32 / 8)
constexpr int CH = 128;           // keys per inner chunk
constexpr int NR = CH / 16;       // K (and V) rows per lane per chunk
constexpr int PW = G * D + 32;    // floats per partial record: o[4][128] | m[4] l[4] pad  (17 x 128-byte lines)

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

struct Shared {
    __attribute__((aligned(16))) float q_s[G][D];          // rotated, pre-scaled queries
    __attribute__((aligned(16))) bf16_t knew[D], vnew[D];  // the new token's (rotated) K and V
    __attribute__((aligned(16))) float sc[G][CH];          // scores -> probabilities of the chunk
    float stat[3 * G];                                     // per head: alpha | m_run | l_run
    unsigned last_flag;
    __attribute__((aligned(16))) float red[16][G][D];      // PV partial sums per key slot (32 KiB)
};

// `row` = the new token's projection [q heads | k heads | v heads] x D: global memory (stand-alone kernel) or an LDS copy the caller
// gathered (persistent kernel, OUT_SC1: the merged heads are also stored write-through, packed two bf16 per 4-byte store, for the
// in-launch hand-off to the O projection).  Returns true in the workgroup that merged (the last arriver of its KV head).
// LEAN (persistent kernel: 12-wave workgroups, 168 registers per lane): the rotated queries are re-read from LDS for every key row
// instead of living in 32 registers - same values, same order of operations.
template <bool OUT_SC1, bool LEAN = false>
__device__ __forceinline__ bool body(const FusedParams& p, const bf16_t* row, int split, int hk, int b, int tid, bool act, Shared& sm) {
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane & 15, ks = (wave & 3) * 4 + (lane >> 4);     // 8-wide d chunk, key slot 0..15
    const int slot = p.slot[b], len = slot + 1, position = p.pos[b];
    const int lo = p.window > 0 ? max(0, len - p.window) : 0;
    const int span = len - lo;
    // a split takes at least one chunk: a short context (a 16-row batch at ctx 100) is served by ONE workgroup per KV head
    // and row instead of nsplit mostly empty ones all running the ticket protocol; the others leave at once
    const int per = max(CH, (span + p.nsplit - 1) / p.nsplit);
    const int n_active = (span + per - 1) / per;
    if (split >= n_active) return false;
    const int k0 = lo + split * per, k1 = min(len, k0 + per);
    bf16_t* kb = p.kc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;
    bf16_t* vb = p.vc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * D;

    // ---- the first chunk's K and V rows go in flight before anything else (the new token's row, index `slot`,
    //      is not in the cache yet: it is patched in from LDS after the RoPE) ----
    uint4 kreg[NR], vreg[NR];
    auto issue_k = [&](int c0, int n) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            kreg[i] = make_uint4(0, 0, 0, 0);
            if (act && kk < n && kidx != slot) kreg[i] = *(const uint4*)(kb + (size_t)kidx * D + sub * 8);
        }
    };
    auto issue_v = [&](int c0, int n) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int kk = ks + 16 * i, kidx = c0 + kk;
            vreg[i] = make_uint4(0, 0, 0, 0);
            if (act && kk < n && kidx != slot) vreg[i] = *(const uint4*)(vb + (size_t)kidx * D + sub * 8);
        }
    };
    if (k0 < k1) { issue_k(k0, min(CH, k1 - k0)); issue_v(k0, min(CH, k1 - k0)); }

    // ---- RoPE: thread (h = wave, pair = lane) rotates (d, d+64) of query head hk*G + h ----
    if (act) {
        const float c = p.cosT[(size_t)position * (D / 2) + lane], s = p.sinT[(size_t)position * (D / 2) + lane];
        const bf16_t* qh = row + (size_t)(hk * G + wave) * D;
        const float x = bf16_to_f32(qh[lane]), y = bf16_to_f32(qh[lane + 64]);
        // rounded to bf16 exactly like the stand-alone RoPE kernel before the attention consumes it
        sm.q_s[wave][lane] = bf16_to_f32(f32_to_bf16(x * c - y * s)) * p.scale;
        sm.q_s[wave][lane + 64] = bf16_to_f32(f32_to_bf16(y * c + x * s)) * p.scale;
        if (wave == 0) {
            const bf16_t* kh = row + (size_t)(p.Hq + hk) * D;
            const float kx = bf16_to_f32(kh[lane]), ky = bf16_to_f32(kh[lane + 64]);
            sm.knew[lane] = f32_to_bf16(kx * c - ky * s);
            sm.knew[lane + 64] = f32_to_bf16(ky * c + kx * s);
        } else if (wave == 1) {
            const bf16_t* vh = row + (size_t)(p.Hq + p.Hkv + hk) * D;
            sm.vnew[lane] = vh[lane];
            sm.vnew[lane + 64] = vh[lane + 64];
        }
    }
    if (act && tid < G) { sm.stat[G + tid] = -INFINITY; sm.stat[2 * G + tid] = 0.f; }
    __syncthreads();
    if (act && split == 0 && tid < 32) {   // one workgroup per (slot, kv head) appends the new row to the cache
        if (tid < 16) *(uint4*)(kb + (size_t)slot * D + tid * 8) = *(const uint4*)(sm.knew + tid * 8);
        else *(uint4*)(vb + (size_t)slot * D + (tid - 16) * 8) = *(const uint4*)(sm.vnew + (tid - 16) * 8);
    }

    float qr[G][8];
    float acc[G][8];
#pragma unroll
    for (int h = 0; h < G; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) { qr[h][j] = (act && !LEAN) ? sm.q_s[h][sub * 8 + j] : 0.f; acc[h][j] = 0.f; }

    for (int c0 = k0; c0 < k1; c0 += CH) {
        const int n = min(CH, k1 - c0);
        // (a split of several chunks - many rows, one split per context: the NEXT chunk's K rows are requested as soon as this chunk's scores have
        // consumed theirs, its V rows after this chunk's P V: the loads pass under the softmax / P V / next scores instead of heading each chunk)
        const int c1 = c0 + CH, n1 = min(CH, k1 - c1);
        if (act) {
            if (slot >= c0 && slot < c0 + n && ((slot - c0) & 15) == ks) {   // this lane group holds the new token's row
                const int i_new = (slot - c0) >> 4;
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (i == i_new) { kreg[i] = *(const uint4*)(sm.knew + sub * 8); vreg[i] = *(const uint4*)(sm.vnew + sub * 8); }
            }
            // ---- scores ----
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int kk = ks + 16 * i;
                const u16x8 kv = __builtin_bit_cast(u16x8, kreg[i]);
                float s[G] = {0.f, 0.f, 0.f, 0.f};
                if (LEAN) {
                    asm volatile("" ::: "memory");       // (keeps the q reads inside the loop: no 32-register copy of q)
#pragma unroll
                    for (int h = 0; h < G; ++h) {
                        const f32x4 qa = *(const f32x4*)&sm.q_s[h][sub * 8], qb = *(const f32x4*)&sm.q_s[h][sub * 8 + 4];
#pragma unroll
                        for (int j = 0; j < 8; ++j) qr[h][j] = j < 4 ? qa[j] : qb[j - 4];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float kf = bf16_to_f32(kv[j]);
#pragma unroll
                    for (int h = 0; h < G; ++h) s[h] += qr[h][j] * kf;
                }
#pragma unroll
                for (int h = 0; h < G; ++h) {
                    // the 16 lanes of a row hold the 16 d-chunks of one key: row-wide sum by four DPP adds on the VALU (quad swaps, half-row mirror,
                    // row mirror; every lane ends with the row's sum).  The xor butterfly compiled to 128 ds_bpermute_b32 per thread and chunk -
                    // four dependent round trips through the LDS pipeline per (key, head)
                    s[h] = dpp_add<0xb1, 0xf>(s[h]); s[h] = dpp_add<0x4e, 0xf>(s[h]);
                    s[h] = dpp_add<0x141, 0xf>(s[h]); s[h] = dpp_add<0x140, 0xf>(s[h]);
                }
                if (sub == 0 && kk < n) {
#pragma unroll
                    for (int h = 0; h < G; ++h) sm.sc[h][kk] = s[h];
                }
            }
        }
        if (c1 < k1) issue_k(c1, n1);
        __syncthreads();
        // ---- online softmax, wave h owns head h ----
        if (act) {
            const float s0 = lane < n ? sm.sc[wave][lane] : -INFINITY;
            const float s1 = lane + 64 < n ? sm.sc[wave][lane + 64] : -INFINITY;
            const float m_old = sm.stat[G + wave];
            const float m_new = fmaxf(m_old, wave_max(fmaxf(s0, s1)));
            const float e0 = __expf(s0 - m_new), e1 = __expf(s1 - m_new);
            if (lane < n) sm.sc[wave][lane] = e0;
            if (lane + 64 < n) sm.sc[wave][lane + 64] = e1;
            const float ps = wave_sum(e0 + e1);
            const float alpha = __expf(m_old - m_new);
            if (lane == 0) { sm.stat[wave] = alpha; sm.stat[G + wave] = m_new; sm.stat[2 * G + wave] = sm.stat[2 * G + wave] * alpha + ps; }
        }
        __syncthreads();
        // ---- O += P V ----
        if (act) {
            float al[G];
#pragma unroll
            for (int h = 0; h < G; ++h) al[h] = sm.stat[h];
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
                    for (int h = 0; h < G; ++h) pr[h] = sm.sc[h][kk];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float vf = bf16_to_f32(vv[j]);
#pragma unroll
                        for (int h = 0; h < G; ++h) acc[h][j] += pr[h] * vf;
                    }
                }
            }
        }
        if (c1 < k1) issue_v(c1, n1);
        __syncthreads();   // sc / stat are rewritten by the next chunk
    }
    // ---- reduce the 16 key slots through LDS, write this split's partial (write-through stores) ----
    if (act) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
            *(f32x4*)&sm.red[ks][h][sub * 8] = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
            *(f32x4*)&sm.red[ks][h][sub * 8 + 4] = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
        }
    }
    __syncthreads();
    if (n_active == 1) {
        // ONE split covers the row's whole context (short contexts; wide batches, whose rows fill the chip by themselves): nothing to meet -
        // no partial record, no ticket, no re-read.  The merge below with a single partial multiplies by exp(0) = 1 and divides by the same
        // l: these are the same bits.
        if (act) {
            for (int i = tid; i < G * D; i += 256) {
                const int h = i >> 7, d = i & 127;
                float v = 0.f;
#pragma unroll
                for (int s16 = 0; s16 < 16; ++s16) v += sm.red[s16][h][d];
                const float l = sm.stat[2 * G + h];
                const unsigned short ob = f32_to_bf16(v * (l > 0.f ? 1.0f / l : 0.f));
                bf16_t* dst = p.o + ((size_t)b * p.Hq + hk * G) * D + i;
                if (OUT_SC1) {
                    const unsigned other = __shfl_xor((unsigned)ob, 1, 64);
                    if ((lane & 1) == 0) __hip_atomic_store((unsigned*)dst, (unsigned)ob | (other << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *dst = ob;
                }
            }
        }
        return true;
    }
    float* po = p.part + (((size_t)b * p.Hkv + hk) * p.nsplit + split) * PW;
    if (act) {
        for (int i = tid; i < G * D; i += 256) {
            const int h = i >> 7, d = i & 127;
            float v = 0.f;
#pragma unroll
            for (int s16 = 0; s16 < 16; ++s16) v += sm.red[s16][h][d];
            st_sc1(po + i, v);
        }
        if (tid < G) { st_sc1(po + G * D + tid, sm.stat[G + tid]); st_sc1(po + G * D + G + tid, sm.stat[2 * G + tid]); }
        // ---- every storing wave drains its stores, then ONE lane takes the ticket; the last arriver merges ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.ticket + (size_t)b * p.Hkv + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sm.last_flag = (t == (unsigned)n_active - 1) ? 1u : 0u;
    }
    __syncthreads();   // the wave that added joins this barrier after its add returned; everyone loads behind it
    if (!sm.last_flag) return false;
    const float* pp = p.part + ((size_t)b * p.Hkv + hk) * p.nsplit * PW;
    float* wgt = &sm.red[0][0][0];            // [G][64] weights, then [G] 1/l   (red is free again)
    if (act) {
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
    if (act) {
        for (int i = tid; i < G * D; i += 256) {
            const int h = i >> 7;
            float a = 0.f;
#pragma unroll 8
            for (int s2 = 0; s2 < n_active; ++s2) a += wgt[h * 64 + s2] * ld_sc1(pp + (size_t)s2 * PW + i);
            const unsigned short ob = f32_to_bf16(a * wgt[G * 64 + h]);
            bf16_t* dst = p.o + ((size_t)b * p.Hq + hk * G) * D + i;
            if (OUT_SC1) {       // two neighbouring outputs per 4-byte write-through store (the even lane stores the pair)
                const unsigned other = __shfl_xor((unsigned)ob, 1, 64);
                if ((lane & 1) == 0) __hip_atomic_store((unsigned*)dst, (unsigned)ob | (other << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *dst = ob;
            }
        }
    }
    if (tid == 0) __hip_atomic_store(p.ticket + (size_t)b * p.Hkv + hk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// The barrier skeleton of body() for waves of a larger workgroup that take no part in the attention (the persistent kernel's streaming
// waves, which keep weight loads in flight meanwhile): the same uniform control flow, every __syncthreads of body() in the same order,
// no memory traffic except the reads that steer it.  Kept next to body(): a barrier added there must be added here.
__device__ __forceinline__ void shadow(const FusedParams& p, int split, int slot, Shared& sm) {     // slot = p.slot[b], read once by the caller
    const int len = slot + 1;
    const int lo = p.window > 0 ? max(0, len - p.window) : 0;
    const int span = len - lo;
    const int per = max(CH, (span + p.nsplit - 1) / p.nsplit);
    const int n_active = (span + per - 1) / per;
    if (split >= n_active) return;
    const int k0 = lo + split * per, k1 = min(len, k0 + per);
    __syncthreads();                       // RoPE done
    for (int c0 = k0; c0 < k1; c0 += CH) {
        __syncthreads();                   // scores
        __syncthreads();                   // softmax
        __syncthreads();                   // P V
    }
    __syncthreads();                       // key-slot partials in LDS
    if (n_active == 1) return;             // (the direct path: no record, no ticket)
    __syncthreads();                       // partial record stored
    __syncthreads();                       // ticket taken
    if (!sm.last_flag) return;
    __syncthreads();                       // merge weights
}

}  // namespace attn_dec
