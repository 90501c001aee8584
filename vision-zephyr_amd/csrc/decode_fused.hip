// One launch for the attention half of a batch-1 decode layer (gfx950):
//     qkv = W_qkv . rmsnorm(x)            (GEMV, gemv.hip's batch-1 form)
//     o   = attention(RoPE(q), cache + RoPE(k), v)   (attn_decode.hip's fused form: RoPE + append + split attention + merge)
//     x  += W_o . o                       (GEMV with residual)
// Three kernels of 8-16 us each, every one a short HBM burst behind a dependent chain of memory round trips, spend a large part
// of their time ramping up and draining (QKV / O run at 4.0-4.7 TB/s where the long GEMVs reach 5.6-6.3, the
// attention moves 8 MB in 16 us).  Here the three are ROLES of one grid, ordered [attention | QKV rows | O rows] by block index:
//   * an attention workgroup requests its K/V rows first and only then waits for the QKV rows of its KV head (48 producer
//     workgroups of 16 rows: 32 of q, 8 of k, 8 of v); the cache streams in under the QKV GEMV instead of after it;
//   * an O-projection wave requests ALL of its weights (4 rows x 8 KiB, registers) once the QKV rows are in and only then waits
//     for the eight head merges; by the time the attention is done its weights have landed - the projection costs a dot product;
//   * two kernel boundaries (1.2-1.9 us each, MI355X_MICROARCH "boundary") disappear.
// Hand-off (CDNA4 guide section 6 G16, flag form): the producer's payload goes out with sc1 (write-through) stores, every storing
// wave drains its stores (vmcnt(0)), a workgroup barrier, then ONE lane stores the EPOCH into the workgroup's own flag word (sc1);
// wave 0 of a consumer polls the flag words of its producers (one or two per lane, sc1 loads, bounded), a workgroup barrier, then
// everybody loads the payload with sc1 loads.  epoch = step + 1, `step` being the device-side token counter of vz_llm_decode_steps
// (the host zeroes the flags and step together before the first launch): flags only grow, nothing is reset inside the launch.
// Forward progress: waiting workgroups (attention: nsplit x 8; O rows: 256) only ever wait for workgroups with a LOWER role
// position in the dependency chain that are already dispatched: the attention blocks come first in the grid and are fewer than the
// resident capacity (checked on the host with the occupancy API, otherwise the engine keeps the three-kernel path); the O blocks
// come last, after every producer has been dispatched.  Every poll loop is bounded: on expiry the workgroup raises the engine's
// error word and goes on (garbage out, no hang); the host checks the word when it reads the ids.
// Arithmetic is the unfused kernels' exactly (same accumulation order everywhere): ids and logits are bit-identical (tested).
//
// STATUS (round 1, tools/fused_stamps.py, ctx 2048): the launch takes 33 us against 36.5 us for the three kernels (12.6 + 16 +
// 8.4) and the decode rate is unchanged within noise (329 vs 332 tok/s), so the engine keeps the three kernels unless
// vz_tune_set(12, 1).  Timeline: QKV rows published 8 .. 15-18 us (standalone GEMV: 12.6), attention chain after its last
// QKV row 9.5 us (RoPE .. split partials .. merge), O rows 3.5 us after the last merge.  What was learnt:
//   * arrival COUNTERS (agent-scope atomic adds, 768 producers + 650 pollers on one 128-byte line) serialise at the memory side:
//     66 us; one flag word per producer polled by a wave: 33 us;
//   * the kernel needs 168 VGPRs (the attention role) = 3 workgroups per CU: with 2 rows per wave the 768 QKV workgroups need a
//     second residency round; with 4 rows per wave (all 32 KiB per wave in flight) 384 workgroups sit 1 or 2 to a CU and the CUs
//     with two finish last - a CU pulls ~24 GB/s whatever it has in flight - so the QKV phase ends at 15 us either way;
//   * letting the O rows request their weights from t = 0 costs the QKV rows 6 us (HBM order), gating them on "QKV rows in"
//     moves the same bytes behind the QKV stream but they then race the slowest heads.
// Next step: weights through an LDS-DMA ring per CU (one loader wave, equal bytes per CU by construction) instead of per-wave
// register batches - the guide's batch-1 engine (cdna_hip_programming.md section 5.6).
#include "vz_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int D = 128, G = 4, CH = 128, NR = CH / 16, PW = G * D + 32;
constexpr long SPIN_LIMIT = 1L << 22;        // polls of >= 256 clocks each: seconds, far beyond any legitimate wait

struct FusedLayerParams {
    // residual stream (1 row) and the three weights
    bf16_t* x;                         // [H] in: layer input; out: x + W_o . o
    const float* norm_w; float norm_eps;
    const bf16_t* Wqkv; const unsigned char* Wqkv8; const float* sqkv;     // [QKV][H]
    const bf16_t* Wo; const unsigned char* Wo8; const float* so;           // [H][A]
    bf16_t* qkv;                       // [QKV] scratch (sc1 traffic)
    bf16_t* att;                       // [A] scratch (sc1 traffic)
    // attention
    bf16_t *kc, *vc; float* part; unsigned* split_ticket;
    const float *cosT, *sinT; const int *pos, *slot;
    int H, QKV, A, Hq, Hkv, max_ctx, nsplit, window;
    float scale;
    // hand-off flags: tq[QKV / 8] (one per QKV-row workgroup), to[Hkv] (head merged); step = device token counter
    unsigned* tq; unsigned* to; const int* step; int* err;
    int nB, nA, nC;                    // role sizes in workgroups
    long long* stamps;                 // profiling (knob 13): 4 x s_memrealtime per workgroup: start | wait done | finished | -
};

__device__ __forceinline__ void st_sc1_u32(void* p, unsigned v) { __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_sc1_u32(const void* p) { return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { st_sc1_u32(p, __float_as_uint(v)); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __uint_as_float(ld_sc1_u32(p)); }

// Hand-off flags instead of counters: every producer workgroup owns one word and stores the epoch into it (sc1, after its payload
// has drained); wave 0 of a consumer polls the words of ITS producers, one or two per lane, and everyone leaves through the
// barrier behind it.  (The first version counted arrivals with agent-scope atomic adds: 768 adds + 650 pollers on one 128-byte
// line serialised at the memory side - the launch took 66 us instead of the 40 us of the three kernels it replaced.)
// idx0 / idx1: this lane's flag indices, -1 = none.
__device__ __forceinline__ void wait_flags(const unsigned* flags, int idx0, int idx1, unsigned epoch, int* err, int sleep) {
    if (threadIdx.x < 64) {
        long spins = 0;
        for (;;) {
            bool ok = true;
            if (idx0 >= 0) ok = ld_sc1_u32(flags + idx0) >= epoch;
            if (idx1 >= 0) ok = ok && ld_sc1_u32(flags + idx1) >= epoch;
            if (__all(ok)) break;
            if (sleep) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(4);
            if (++spins > SPIN_LIMIT) { if (threadIdx.x == 0) *err = 1; break; }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w0), __builtin_bit_cast(bf16x2, x0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w1), __builtin_bit_cast(bf16x2, x1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w2), __builtin_bit_cast(bf16x2, x2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w3), __builtin_bit_cast(bf16x2, x3), acc, false);
    return acc;
}
__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned w, bool hi_half) {
    return hi_half ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                   : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
}
__device__ __forceinline__ void fp8x16_to_bf16(const u32x4 w, u32x4& lo, u32x4& hi) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lo = (u32x4){fp8x2_to_bf16x2(w0, false), fp8x2_to_bf16x2(w0, true), fp8x2_to_bf16x2(w1, false), fp8x2_to_bf16x2(w1, true)};
    hi = (u32x4){fp8x2_to_bf16x2(w2, false), fp8x2_to_bf16x2(w2, true), fp8x2_to_bf16x2(w3, false), fp8x2_to_bf16x2(w3, true)};
}

// ---- batch-1 GEMV of one workgroup: 4 waves x R = 4 weight rows (rows 16*blk .. 16*blk+15), every chunk of every row in flight
// at once (4 x 8 KiB per wave, registers).  Per row the arithmetic of gemv_bf16_kernel<1, 2, 8, true, FP8, 4>: per lane a dot8
// chain over the chunks in k order, then wave_sum_lane63.  issue() requests the weights, finish() - once the activations are
// in LDS (`xs`, K bf16) - does the dot products.
template <bool FP8>
struct GemvRows {
    static constexpr int R = 4, EPL = FP8 ? 16 : 8, CHK = 64 * EPL, MAXC = FP8 ? 4 : 8;     // K <= 4096
    u32x4 w[R][MAXC];
    int nchunk;
    __device__ __forceinline__ void issue(const bf16_t* W, const unsigned char* W8, int ldw, int K, int row0, int lane) {
        nchunk = K / CHK;
        const char* base = FP8 ? (const char*)W8 : (const char*)W;
        const int wb = FP8 ? 1 : 2;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const char* wp = base + ((size_t)(row0 + r) * ldw + lane * EPL) * wb;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < nchunk) w[r][c] = __builtin_nontemporal_load((const u32x4*)(wp + (size_t)c * CHK * wb));
        }
    }
    __device__ __forceinline__ void finish(const bf16_t* xs, int lane, float (&a)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nchunk) {
                if constexpr (FP8) {
                    const bf16_t* xp = xs + c * CHK + lane * 16;
                    const u32x4 x0 = *(const u32x4*)xp, x1 = *(const u32x4*)(xp + 8);
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        u32x4 lo, hi;
                        fp8x16_to_bf16(w[r][c], lo, hi);
                        a[r] = dot8(hi, x1, dot8(lo, x0, a[r]));
                    }
                } else {
                    const u32x4 xv = *(const u32x4*)(xs + c * 512 + lane * 8);
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] = dot8(w[r][c], xv, a[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = wave_sum_lane63(a[r]);
    }
};

template <bool FP8>
__global__ __launch_bounds__(256, 3) void decode_attn_half_kernel(FusedLayerParams p) {
    // LDS: the roles never coexist in a workgroup; the GEMV roles use xs (<= 8192 bf16 = 16 KiB) + red, the attention its own arrays
    __shared__ __attribute__((aligned(16))) char lds_raw[16 * G * D * 4 + 2 * G * D * 4 + 4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x;
    const unsigned epoch = (unsigned)(*p.step) + 1u;
#define FSTAMP(i) if (p.stamps && tid == 0) p.stamps[(size_t)bid * 4 + (i)] = (long long)__builtin_amdgcn_s_memrealtime();
    FSTAMP(0)

    if (bid >= p.nB) {
        // =====================================================================================================
        // GEMV roles: QKV rows (bid - nB < nA) or O rows
        // =====================================================================================================
        const bool is_qkv = bid - p.nB < p.nA;
        const int blk = is_qkv ? bid - p.nB : bid - p.nB - p.nA;
        const int K = is_qkv ? p.H : p.A;
        const int row0 = blk * 16 + wave * 4;
        bf16_t* xs = (bf16_t*)lds_raw;
        float* red = (float*)(lds_raw + (size_t)K * 2);
        GemvRows<FP8> gv;
        if (is_qkv) {
            gv.issue(p.Wqkv, p.Wqkv8, p.H, K, row0, lane);
        } else {
            // HBM order: the QKV weights are needed first, the O weights last (after the attention chain, ~9 us): an O workgroup
            // holds its requests back until the attention workgroup of ONE KV head (its index mod Hkv) has seen all of that head's
            // QKV rows - the QKV stream is through by then - instead of competing with it from t = 0 (measured: QKV rows done at
            // 18 us with the competition, 12 us without); the 33 MB then stream under the attention chain.
            if (p.nA > 0) wait_flags(p.to + 32, lane == 0 ? blk % p.Hkv : -1, -1, epoch, p.err, 1);
            gv.issue(p.Wo, p.Wo8, p.A, K, row0, lane);
        }

        if (is_qkv) {
            // RMSNorm of the residual stream fused into the staging (gemv.hip, MB = 1 form: same reduction order)
            const bf16_t* x = p.x;
            float ss = 0.f;
            for (int k = tid * 8; k < K; k += 256 * 8) {
                const u16x8 v = *(const u16x8*)(x + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(v[j]); ss += f * f; }
            }
            ss = wave_sum(ss);
            __syncthreads();
            if (lane == 0) red[wave] = ss;
            __syncthreads();
            float tot = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) tot += red[w2];
            const float rstd = rsqrtf(tot / (float)K + p.norm_eps);
            for (int k = tid * 8; k < K; k += 256 * 8) {
                const u16x8 v = *(const u16x8*)(x + k);
                const f32x4 w0 = *(const f32x4*)(p.norm_w + k), w1 = *(const f32x4*)(p.norm_w + k + 4);
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float wj = j < 4 ? w0[j] : w1[j - 4];
                    o[j] = f32_to_bf16(wj * (bf16_to_f32(v[j]) * rstd));
                }
                *(u16x8*)(xs + k) = o;
            }
            __syncthreads();
            float a[4];
            gv.finish(xs, lane, a);
            if (lane == 63) {
                if constexpr (FP8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= p.sqkv[row0 + r];
                }
                st_sc1_u32(p.qkv + row0, (unsigned)f32_to_bf16(a[0]) | ((unsigned)f32_to_bf16(a[1]) << 16));
                st_sc1_u32(p.qkv + row0 + 2, (unsigned)f32_to_bf16(a[2]) | ((unsigned)f32_to_bf16(a[3]) << 16));
            }
            // publish: this workgroup's 16 rows belong to one KV head (16 | 128)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) st_sc1_u32(p.tq + blk, epoch);          // this workgroup's 8 rows are out
            FSTAMP(2)
        } else {
            // O rows: weights are in flight; wait for the eight head merges, stage o, finish
            wait_flags(p.to, lane < p.Hkv ? lane : -1, -1, epoch, p.err, 1);
            FSTAMP(1)
            for (int k = tid * 2; k < K; k += 256 * 2) *(unsigned*)(xs + k) = ld_sc1_u32(p.att + k);
            __syncthreads();
            float a[4];
            gv.finish(xs, lane, a);
            if (lane == 63) {
                if constexpr (FP8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= p.so[row0 + r];
                }
                const uint2 rx = *(const uint2*)(p.x + row0);                       // residual: the layer input (nobody reads x any more)
                a[0] += bf16_to_f32((bf16_t)(rx.x & 0xffffu)); a[1] += bf16_to_f32((bf16_t)(rx.x >> 16));
                a[2] += bf16_to_f32((bf16_t)(rx.y & 0xffffu)); a[3] += bf16_to_f32((bf16_t)(rx.y >> 16));
                uint2 o;
                o.x = (unsigned)f32_to_bf16(a[0]) | ((unsigned)f32_to_bf16(a[1]) << 16);
                o.y = (unsigned)f32_to_bf16(a[2]) | ((unsigned)f32_to_bf16(a[3]) << 16);
                *(uint2*)(p.x + row0) = o;
            }
            FSTAMP(2)
        }
        return;
    }

    // =========================================================================================================
    // attention role: attn_decode_fused_kernel's body for batch row 0, with the wait for its KV head's QKV rows placed
    // AFTER the first chunk's K/V requests
    // =========================================================================================================
    float (*q_s)[D] = (float (*)[D])lds_raw;                                   // [G][D]
    bf16_t* knew = (bf16_t*)(lds_raw + G * D * 4);                             // [D]
    bf16_t* vnew = knew + D;                                                   // [D]
    float (*sc)[CH] = (float (*)[CH])(lds_raw + G * D * 4 + 2 * D * 2);        // [G][CH]
    float* stat = (float*)(lds_raw + G * D * 4 + 2 * D * 2 + G * CH * 4);      // [3 G]
    unsigned* last_flag = (unsigned*)(stat + 3 * G);
    float (*red)[G][D] = (float (*)[G][D])(lds_raw + 2 * G * D * 4 + 4096);    // [16][G][D]

    const int split = bid % p.nsplit, hk = bid / p.nsplit;
    const int sub = lane & 15, ks = wave * 4 + (lane >> 4);
    const int slot = p.slot[0], len = slot + 1, position = p.pos[0];
    const int lo = p.window > 0 ? max(0, len - p.window) : 0;
    const int span = len - lo;
    const int per = max(CH, (span + p.nsplit - 1) / p.nsplit);
    const int n_active = (span + per - 1) / per;
    if (split >= n_active) return;
    const int k0 = lo + split * per, k1 = min(len, k0 + per);
    const int heads = p.Hq + 2 * p.Hkv;
    (void)heads;
    const bf16_t* row = p.qkv;
    bf16_t* kb = p.kc + (size_t)hk * (size_t)p.max_ctx * D;
    bf16_t* vb = p.vc + (size_t)hk * (size_t)p.max_ctx * D;

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

    // ---- the new token's q / k / v rows of this KV head: 96 producer workgroups ----
    if (p.nA > 0) {
        // producers of this KV head: 32 workgroups of q rows, 8 of k rows, 8 of v rows (16 rows each)
        const int qb = p.Hq * D / 16, kb16 = p.Hkv * D / 16;
        const int i0 = lane < 32 ? hk * (G * D / 16) + lane : -1;
        const int i1 = lane < 8 ? qb + hk * (D / 16) + lane : (lane < 16 ? qb + kb16 + hk * (D / 16) + (lane - 8) : -1);
        wait_flags(p.tq, i0, i1, epoch, p.err, 0);
        if (split == 0 && tid == 0) st_sc1_u32(p.to + 32 + hk, epoch);      // "this head's QKV rows are in": releases the O rows' weight requests
    }
    FSTAMP(1)
    {
        const float c = p.cosT[(size_t)position * (D / 2) + lane], s = p.sinT[(size_t)position * (D / 2) + lane];
        auto ld16 = [&](const bf16_t* base, int i) -> float {
            const unsigned u = ld_sc1_u32(base + (i & ~1));
            return bf16_to_f32((bf16_t)((i & 1) ? (u >> 16) : (u & 0xffffu)));
        };
        const bf16_t* qh = row + (size_t)(hk * G + wave) * D;
        const float x = ld16(qh, lane), y = ld16(qh, lane + 64);
        q_s[wave][lane] = bf16_to_f32(f32_to_bf16(x * c - y * s)) * p.scale;
        q_s[wave][lane + 64] = bf16_to_f32(f32_to_bf16(y * c + x * s)) * p.scale;
        if (wave == 0) {
            const bf16_t* kh = row + (size_t)(p.Hq + hk) * D;
            const float kx = ld16(kh, lane), ky = ld16(kh, lane + 64);
            knew[lane] = f32_to_bf16(kx * c - ky * s);
            knew[lane + 64] = f32_to_bf16(ky * c + kx * s);
        } else if (wave == 1) {
            const bf16_t* vh = row + (size_t)(p.Hq + p.Hkv + hk) * D;
            vnew[lane] = f32_to_bf16(ld16(vh, lane));
            vnew[lane + 64] = f32_to_bf16(ld16(vh, lane + 64));
        }
    }
    if (tid < G) { stat[G + tid] = -INFINITY; stat[2 * G + tid] = 0.f; }
    __syncthreads();
    if (split == 0 && tid < 32) {
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
        if (slot >= c0 && slot < c0 + n && ((slot - c0) & 15) == ks) {
            const int i_new = (slot - c0) >> 4;
#pragma unroll
            for (int i = 0; i < NR; ++i)
                if (i == i_new) { kreg[i] = *(const uint4*)(knew + sub * 8); vreg[i] = *(const uint4*)(vnew + sub * 8); }
        }
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
        __syncthreads();
    }
#pragma unroll
    for (int h = 0; h < G; ++h) {
        *(f32x4*)&red[ks][h][sub * 8] = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
        *(f32x4*)&red[ks][h][sub * 8 + 4] = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
    }
    __syncthreads();
    float* po = p.part + ((size_t)hk * p.nsplit + split) * PW;
    for (int i = tid; i < G * D; i += 256) {
        const int h = i >> 7, d = i & 127;
        float v = 0.f;
#pragma unroll
        for (int s16 = 0; s16 < 16; ++s16) v += red[s16][h][d];
        st_sc1(po + i, v);
    }
    if (tid < G) { st_sc1(po + G * D + tid, stat[G + tid]); st_sc1(po + G * D + G + tid, stat[2 * G + tid]); }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.split_ticket + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_flag = (t == (unsigned)n_active - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!*last_flag) { FSTAMP(3) return; }
    const float* pp = p.part + (size_t)hk * p.nsplit * PW;
    float* wgt = &red[0][0][0];
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
    {
        // two adjacent outputs per thread (256 x 2 = G x D): one 4-byte write-through store each
        const int i = tid * 2, h = i >> 7;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
        for (int s2 = 0; s2 < n_active; ++s2) {
            const float w = wgt[h * 64 + s2];
            a0 += w * ld_sc1(pp + (size_t)s2 * PW + i);
            a1 += w * ld_sc1(pp + (size_t)s2 * PW + i + 1);
        }
        const float inv = wgt[G * 64 + h];
        st_sc1_u32(p.att + (size_t)hk * G * D + i, (unsigned)f32_to_bf16(a0 * inv) | ((unsigned)f32_to_bf16(a1 * inv) << 16));
    }
    if (tid == 0) __hip_atomic_store(p.split_ticket + hk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // publish this head's 512 outputs to the O rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) st_sc1_u32(p.to + hk, epoch);
    FSTAMP(2)
}

long long* g_fstamps = nullptr;   // 4 stamps per workgroup of the last stamped launch
int g_fstamp_wgs = 0;
int g_capacity[2] = {0, 0};     // resident workgroups of the two instantiations (occupancy API x CUs), 0 = not queried yet

}  // namespace

int vz_attn_half_capacity(bool fp8);
int g_decode_fuse_stamps = 0;   // vz_tune_set(13, v): record role stamps of the NEXT launches (tools/fused_stamps.py)
int g_decode_fuse = 0;          // vz_tune_set(12, v): 1 = one launch for QKV GEMV + attention + O GEMV of a batch-1 decode layer.
                                // OFF by default: correct (bit-identical, tested) but not faster yet - see STATUS below

bool vz_attn_half_ok(const AttnHalfArgs& a) {
    if (!g_decode_fuse) return false;
    if (a.H % 512 || a.A % 512 || a.H > 4096 || a.A > 4096) return false;      // every chunk of 4 rows in flight per wave: K <= 4096
    if (a.fp8 && (a.H % 1024 || a.A % 1024)) return false;
    if (a.Hq != a.Hkv * G || a.QKV != (a.Hq + 2 * a.Hkv) * D || a.A != a.Hq * D || a.QKV % 16 || a.H % 16) return false;
    if (a.nsplit < 1 || a.nsplit > 64 || a.QKV / 8 > 960 || a.Hkv > 64) return false;                    // flag words per layer
    return true;
}

int vz_launch_attn_half(const AttnHalfArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(vz_attn_half_ok(a), "decode attention half: unsupported shape");
    const int which = a.fp8 ? 1 : 0;
    (void)vz_attn_half_capacity(a.fp8);
    FusedLayerParams p;
    p.x = a.x; p.norm_w = a.norm_w; p.norm_eps = a.norm_eps;
    p.Wqkv = a.Wqkv; p.Wqkv8 = a.Wqkv8; p.sqkv = a.sqkv; p.Wo = a.Wo; p.Wo8 = a.Wo8; p.so = a.so;
    p.qkv = a.qkv; p.att = a.att; p.kc = a.kc; p.vc = a.vc; p.part = a.part; p.split_ticket = a.split_ticket;
    p.cosT = a.cosT; p.sinT = a.sinT; p.pos = a.pos; p.slot = a.slot;
    p.H = a.H; p.QKV = a.QKV; p.A = a.A; p.Hq = a.Hq; p.Hkv = a.Hkv; p.max_ctx = a.max_ctx; p.nsplit = a.nsplit; p.window = a.window;
    p.scale = a.scale; p.tq = a.tq; p.to = a.to; p.step = a.step; p.err = a.err;
    p.nB = a.nsplit * a.Hkv; p.nA = g_decode_fuse == 2 ? 0 : a.QKV / 16; p.nC = a.H / 16;      // mode 2: the QKV GEMV ran as its own launch
    // the waiting attention workgroups must leave room for their producers (see the header): otherwise the caller's three-kernel path
    VZ_CHECK_ARG(g_capacity[which] > 0 && p.nB + 64 <= g_capacity[which], "decode attention half: %d waiting workgroups do not fit %d resident slots",
                 p.nB, g_capacity[which]);
    const dim3 grid(p.nB + p.nA + p.nC);
    p.stamps = nullptr;
    if (g_decode_fuse_stamps) {
        if (!g_fstamps) VZ_CHECK_HIP(hipMalloc((void**)&g_fstamps, (size_t)4096 * 4 * sizeof(long long)));
        if ((int)grid.x <= 4096) { p.stamps = g_fstamps; g_fstamp_wgs = grid.x; VZ_CHECK_HIP(hipMemsetAsync(g_fstamps, 0, (size_t)grid.x * 4 * sizeof(long long), s)); }
    }
    if (a.fp8) vz_launch_timed(decode_attn_half_kernel<true>, grid, dim3(256), 0, s, p);
    else vz_launch_timed(decode_attn_half_kernel<false>, grid, dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

// resident workgroups of the kernel on this device (occupancy API x CUs); -1 if the query fails
int vz_attn_half_capacity(bool fp8) {
    const int which = fp8 ? 1 : 0;
    if (!g_capacity[which]) {
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t er = fp8 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_attn_half_kernel<true>, 256, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_attn_half_kernel<false>, 256, 0);
        if (er == hipSuccess) er = hipGetDevice(&dev);
        if (er == hipSuccess) er = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        g_capacity[which] = (er == hipSuccess && per_cu * cus > 0) ? per_cu * cus : -1;
    }
    return g_capacity[which];
}

// profiling: role stamps of the last stamped launch; host_out[4 * wg + {0 start, 1 wait done, 2 finished / published, 3 early exit}]
int vz_attn_half_read_stamps(long long* host, int max_wgs, int* n_wgs, int* nB, int* nA) {
    VZ_CHECK_ARG(host && n_wgs, "stamps: null argument");
    VZ_CHECK_HIP(hipDeviceSynchronize());
    const int n = g_fstamp_wgs < max_wgs ? g_fstamp_wgs : max_wgs;
    if (n > 0 && g_fstamps) VZ_CHECK_HIP(hipMemcpy(host, g_fstamps, (size_t)n * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    *n_wgs = n;
    (void)nB; (void)nA;
    return VZ_OK;
}
