// 256x256x64 deep-pipelined bf16 MFMA GEMM for gfx950:  C[M,N'] = epi(A[M,K] . W[N,K]^T)
//
// The 128x128 kernel (gemm.hip) needs the full LDS read bandwidth of a CU to feed its MFMAs (64x64 wave
// tiles: 512 B of LDS per 16x16x32 MFMA) and stalls on one vmcnt(0)+barrier per K-tile: ~36-39 % of
// the 2.5 PF peak.  This kernel follows the structure of the CDNA4 guide (section 5, "The 256^2
// 8-phase template"):
//   * 8 waves (2 M x 4 N), each owning a 128x64 output tile = 32 accumulators of 16x16 (384 B of LDS
//     per MFMA), one workgroup per CU.  A half-tile X_h holds the rows every wave needs for its output
//     quadrant h (128 rows x 64 k = 16 KiB).
//   * LDS = all 160 KiB of the CU as a ring of 10 half-tile slots.  The half-tiles of the K-tiles are laid into the
//     ring in the order they are read (A_0, B_0, B_1, A_1 of K-tile t = sequence 4t..4t+3, slot = sequence mod 10)
//     and staged two K-tiles ahead, so 4-6 half-tiles (64-96 KiB per CU) are in flight.
//   * each K-tile is 2 phases of 32 MFMAs (two output quadrants each); measured on MI355X the 4-phase form of the
//     guide spends more time in barrier/LDS latency than in MFMAs with this staging, the 2-phase form is faster:
//        phase alpha: read A_0 (8x ds_read_b128), B_0, B_1 (4x each)  MFMA quadrants (0,0),(0,1)  stage A_0, B_0 of t+2, vmcnt(12)
//        phase beta : read A_1 (8x), B fragments stay in registers    MFMA quadrants (1,0),(1,1)  stage B_1, A_1 of t+2, vmcnt(10)
//     every phase = {ds_reads, 4 LDS-DMA per thread, vmcnt(N), lgkmcnt(0), s_barrier, 32 MFMA, s_barrier}; the two wave
//     groups (upper / lower half of the tile) run one barrier apart, so one group's loads overlap the other's MFMAs.
//   * operands are staged with 16-byte LDS-DMA (global_load_lds_dwordx4) that stays in flight ACROSS the raw
//     s_barriers: the only VMEM waits in the loop are the counted ones.  Hazards:
//       RAW  alpha(t) waits vmcnt(12): everything up to A_1(t) = 4t+3 has landed (only 4t+4..4t+9 are newer); it is read in
//            beta(t), two barriers later (one more than the stagger needs).  beta(t) waits vmcnt(10): up to B_1(t+1) = 4t+6
//            (4t+7..4t+11 newer); read in alpha(t+1), again two barriers later.
//       WAR  a slot is re-staged one phase after its last ds_read (4t+8/9 reuse the slots of B_1/A_1(t-1), 4t+10/11 those
//            of A_0/B_0(t)), and every wave has passed its lgkmcnt(0) and that phase's first barrier by then.
//     K-tiles past the end are clamped to the last one (identical bytes written to free slots), so the loop has no
//     tail variants and the counts are exact in every iteration.
//   * LDS image lane-linear per DMA instruction; XOR swizzle (chunk ^= row & 7) on the SOURCE address and on
//     the ds_read_b128; XCD-aware bijective tile order; same fused epilogues as gemm.hip.
//   * stream-K tail: T tiles on P CUs leave a last partial round (Zephyr gate-up at S=2048: 896 tiles = 3.5 rounds;
//     QKV 192 tiles, O / down 128 tiles = less than one).  Whole tiles of the full rounds run one per workgroup; the
//     K-tiles of the remaining tiles are cut evenly over (up to) P more workgroups.  A workgroup that ends up with a
//     K-slice of a tile takes an arrival ticket; all but the last arriver park their fp32 accumulators in a workspace
//     slot (write-through stores), the last arriver adds them (in slice order when there are more than two, so results
//     do not depend on arrival order) and runs the epilogue.  The only wait is the finisher's for slices whose owners
//     have already arrived and are storing (bounded; no workgroup waits for one that may not be running).
#include <map>
#include <mutex>
#include <utility>

#include "vz_common.h"

namespace {

constexpr int HALF_BYTES = 128 * 64 * 2;       // 16 KiB: 128 rows x 64 k
constexpr int RING_SLOTS = 10;
constexpr int RING_BYTES = RING_SLOTS * HALF_BYTES;   // 160 KiB
constexpr int TILE_FLOATS = 256 * 256;
constexpr int SK_MIN_UNITS = 16;               // shortest K-slice (in K-tiles) worth a workgroup of its own

struct Gemm256Params {
    const bf16_t* A; const bf16_t* W; void* C;
    const float* bias; const bf16_t* residual;
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32, tiles_m, tiles_n;
    int drain;               // experiment knob 11: 1 = every workgroup waits for its epilogue stores before it ends (the old behaviour)
    // stream-K: workgroups [0, n_full) run tiles [0, n_full) whole; the n_rem * nk K-tiles ("units") of the remaining
    // tiles are cut into sk_wgs ranges of units_per_wg
    int n_full, n_rem, sk_wgs, units_per_wg, sk_skew;
    int n_pers;              // workgroups that run the n_full whole tiles: n_full (one tile each) or fewer (persistent: tile b, b + n_pers, ... ; round 3)
    float* ws; int* tickets;
    int* err;                // async error word: a bounded wait of the stream-K fix-up that expired raises VZ_ASYNC_STREAMK here
    long long* stamps;   // profiling only (vz_tune_set(6, 1)): s_memrealtime at phase boundaries, 16 per workgroup
    // FP8 instantiation (gemm256_fp8_kernel): A / W point at e4m3 bytes, lda / ldw / K count bytes = k, a K-tile is 128 k; the
    // fp32 sums are multiplied by ascale[m] * wscale[n] (one power-of-two scale per activation row / weight row) before the epilogue
    const float* ascale; const float* wscale;
};

__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// 8-byte write-through store / L2-bypassing load at agent scope (global_store/load_dwordx2 sc1)
__device__ __forceinline__ void st2_sc1(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x2 ld2_sc1(const float* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (f32x2){__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// one 16x16 accumulator (or a SwiGLU gate/up pair): rows n0..n0+3 of column m
__device__ __forceinline__ void store4(const Gemm256Params& p, int m, int n0, float v[4], int n_out_total, bool vec_ok) {
    if (n0 >= n_out_total) return;
    if (vec_ok && n0 + 3 < n_out_total) {
        if (p.residual) {
            const u16x4 rr = *(const u16x4*)(p.residual + (size_t)m * p.ldr + n0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf16_to_f32(rr[j]);
        }
        if (p.out_fp32) {
            *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (n0 + j >= n_out_total) break;
            float t = v[j];
            if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + j]);
            if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + n0 + j] = t;
            else ((bf16_t*)p.C)[(size_t)m * p.ldc + n0 + j] = f32_to_bf16(t);
        }
    }
}

// rows m = m_base + QM*64 + MT*16 of the wave's output: acc[QM][qn][nt][MT][j] = C[m][n_base + qn*32 + nt*16 + 4g + j]
template <int QM, int MT>
__device__ __forceinline__ void epilogue_rows(const Gemm256Params& p, f32x4 (&acc)[2][2][2][4], int m_base, int n_base, int g,
                                              bool swiglu, int n_out_total, bool vec_ok) {
    const int m = m_base + QM * 64 + MT * 16;
    if (m >= p.M) return;
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        const int nb = n_base + qn * 32;
        if (swiglu) {   // nt 0 = 16 gate rows, nt 1 = the matching 16 up rows
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_silu(acc[QM][qn][0][MT][j]) * acc[QM][qn][1][MT][j];
            store4(p, m, (nb >> 1) + g * 4, v, n_out_total, vec_ok);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n0 = nb + nt * 16 + g * 4;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float tv = acc[QM][qn][nt][MT][j];
                    if (p.bias && n0 + j < p.N) tv += p.bias[n0 + j];
                    v[j] = apply_act(tv, p.act);
                }
                store4(p, m, n0, v, n_out_total, vec_ok);
            }
        }
    }
}

// ---- fast epilogue: all 256 columns of the tile inside N, 8-byte-aligned output rows, 16-byte-aligned bias.
// Everything that is uniform (activation, bias / residual present, output type) is decided once, outside the per-element
// code: the generic path above costs ~30 us per tile in divergent per-element branches, this one ~2 us.
template <int ACT>
__device__ __forceinline__ f32x4 act4(f32x4 v) {
    if constexpr (ACT == VZ_ACT_QUICK_GELU) return (f32x4){act_quick_gelu(v[0]), act_quick_gelu(v[1]), act_quick_gelu(v[2]), act_quick_gelu(v[3])};
    else if constexpr (ACT == VZ_ACT_GELU_ERF) return (f32x4){act_gelu_erf(v[0]), act_gelu_erf(v[1]), act_gelu_erf(v[2]), act_gelu_erf(v[3])};
    else return v;
}

__device__ __forceinline__ void put4(const Gemm256Params& p, bool has_res, bool f32, int m, int n0, f32x4 v) {
    if (has_res) {
        const u16x4 rr = *(const u16x4*)(p.residual + (size_t)m * p.ldr + n0);
        v[0] += bf16_to_f32(rr[0]); v[1] += bf16_to_f32(rr[1]); v[2] += bf16_to_f32(rr[2]); v[3] += bf16_to_f32(rr[3]);
    }
    if (f32) {
        *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = v;
    } else {
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk;
    }
}

template <int ACT, int QM, int MT>
__device__ __forceinline__ void epilogue_fast_rows(const Gemm256Params& p, f32x4 (&acc)[2][2][2][4], const f32x4 (&b4)[2][2],
                                                   int m_base, int n_base, int g, bool has_res, bool f32) {
    const int m = m_base + QM * 64 + MT * 16;
    if (m >= p.M) return;                   // the last tile row may be partial; columns never are on this path
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        if constexpr (ACT == VZ_ACT_SWIGLU) {
            const f32x4 gt = acc[QM][qn][0][MT], up = acc[QM][qn][1][MT];
            const f32x4 v = (f32x4){act_silu(gt[0]) * up[0], act_silu(gt[1]) * up[1], act_silu(gt[2]) * up[2], act_silu(gt[3]) * up[3]};
            put4(p, has_res, f32, m, ((n_base + qn * 32) >> 1) + g * 4, v);
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                put4(p, has_res, f32, m, n_base + qn * 32 + nt * 16 + g * 4, act4<ACT>(acc[QM][qn][nt][MT] + b4[qn][nt]));
        }
    }
}

template <int ACT>
__device__ __forceinline__ void epilogue_fast(const Gemm256Params& p, f32x4 (&acc)[2][2][2][4], int m_base, int n_base, int g) {
    const bool has_res = p.residual != nullptr, f32 = p.out_fp32 != 0;
    f32x4 b4[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            b4[qn][nt] = (ACT != VZ_ACT_SWIGLU && p.bias) ? *(const f32x4*)(p.bias + n_base + qn * 32 + nt * 16 + g * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    epilogue_fast_rows<ACT, 0, 0>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 0, 1>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 0, 2>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 0, 3>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 1, 0>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 1, 1>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 1, 2>(p, acc, b4, m_base, n_base, g, has_res, f32);
    epilogue_fast_rows<ACT, 1, 3>(p, acc, b4, m_base, n_base, g, has_res, f32);
}

// accumulator i of a thread <-> workspace float4 (i * 512 + tid): every wave-instruction moves 1 KiB contiguous
#define VZ_ACC_FOR_EACH(...)                                                              \
    _Pragma("unroll") for (int qm_ = 0; qm_ < 2; ++qm_) {                                 \
        _Pragma("unroll") for (int qn_ = 0; qn_ < 2; ++qn_)                               \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)                               \
        _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) {                             \
            const int i_ = ((qm_ * 2 + qn_) * 2 + nt_) * 4 + mt_;                         \
            f32x4& a_ = acc[qm_][qn_][nt_][mt_];                                          \
            __VA_ARGS__                                                                   \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);   /* 16 accumulators (64 VGPRs of loads in flight) at a time */ \
    }

// acc += slice parked at `src` (sc1 loads).  hipcc keeps atomic loads in program order and waits before each use, so the
// loads of 8 accumulators (16 x 8 bytes per lane) are issued back to back into temporaries first, then added.
#define VZ_ACC_ADD_FROM(src)                                                              \
    _Pragma("unroll") for (int qm_ = 0; qm_ < 2; ++qm_)                                   \
    _Pragma("unroll") for (int qn_ = 0; qn_ < 2; ++qn_) {                                 \
        f32x2 t_[16];                                                                     \
        _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                \
            const int i_ = (qm_ * 2 + qn_) * 8 + e_;                                      \
            t_[2 * e_] = ld2_sc1((src) + (size_t)((i_ * 2) * 512 + tid) * 2);             \
            t_[2 * e_ + 1] = ld2_sc1((src) + (size_t)((i_ * 2 + 1) * 512 + tid) * 2);     \
        }                                                                                 \
        _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                                \
            f32x4& a_ = acc[qm_][qn_][e_ >> 2][e_ & 3];                                   \
            a_[0] += t_[2 * e_][0]; a_[1] += t_[2 * e_][1];                               \
            a_[2] += t_[2 * e_ + 1][0]; a_[3] += t_[2 * e_ + 1][1];                       \
        }                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }

// profiling stamp: 100 MHz constant clock; one lane per workgroup; never read by the kernel
#define VZ_STAMP(idx)                                                                                   \
    if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + (idx)] = (long long)__builtin_amdgcn_s_memrealtime();

__device__ __forceinline__ int ring_adv(int off, int n) {
    off += n * HALF_BYTES;
    return off >= RING_BYTES ? off - RING_BYTES : off;
}

// bijective XCD-aware order: workgroups that share an XCD (bid mod 8) get consecutive work items
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
}

// A phase = {ds_reads, LDS-DMA issue, vmcnt(N), lgkmcnt(0), s_barrier | 32 MFMA, s_barrier}.  The lgkmcnt(0) sits BEFORE the
// first barrier, so once any wave is past that barrier every wave's reads of the phase have completed (the WAR rule
// "re-stage one phase later" then holds by construction, also for the staggered wave group below).
#define PHASE_SYNC_BEGIN()                                  \
    __builtin_amdgcn_sched_barrier(0);                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_setprio(1);
#define PHASE_SYNC_END()                                    \
    __builtin_amdgcn_s_setprio(0);                          \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    asm volatile("" ::: "memory");                          \
    __builtin_amdgcn_sched_barrier(0);

// FP8 = false: bf16 operands, K-tile = 64 k, two k-steps of v_mfma_f32_16x16x32_bf16 per K-tile.  FP8 = true: e4m3 operands, K-tile = 128 k =
// the SAME 128 bytes per row - ring, staging, swizzle, waits and barriers are byte-for-byte those of the bf16 kernel - and ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 2^0) per accumulator and K-tile: the same matrix-pipe cycles per K-tile for twice the
// FLOPs.  A lane's fragment is then 32 contiguous bytes of its row (chunks 2g, 2g+1) instead of chunks g and g+4.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x8_t cat8(bf16x8 lo, bf16x8 hi) {
    const i32x4_t a = __builtin_bit_cast(i32x4_t, lo), b = __builtin_bit_cast(i32x4_t, hi);
    return (i32x8_t){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
template <bool FP8>
__device__ __forceinline__ f32x4 mma2(const bf16x8 (&w)[2], const bf16x8 (&a)[2], f32x4 c) {
    if constexpr (FP8) {
        return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(w[0], w[1]), cat8(a[0], a[1]), c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], a[0], c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], a[1], c, 0, 0, 0);
    }
}

template <bool FP8>
__device__ __forceinline__ void gemm256_body(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY LDS object (a second one makes hipcc drain vmcnt)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;      // 2 x 4 waves, 128(m) x 64(n) each
    const int fr = lane & 15, g = lane >> 4;
    const int nk = FP8 ? p.K >> 7 : p.K >> 6;
    constexpr int ESZ = FP8 ? 1 : 2;
    const int wave_off = wave * 1024;
    const bool late = wm != 0;

    // ---- work: one whole tile, or a range of K-tile units of the remainder tiles ----
    const int bid = blockIdx.x;
    const bool sk = bid >= p.n_pers;
    int u = 0, u_end = nk, tile = 0, jwg = 0;
    int t_lin = bid;                    // whole tiles of this workgroup: t_lin, t_lin + n_pers, ... < n_full
    if (!sk) {
        tile = xcd_order(t_lin, p.n_full);
    } else {
        // range of workgroup j: [j*U + skew(j), (j+1)*U + skew(j+1)), skew = sk_skew for odd j.  Even workgroups get 2*skew
        // K-tiles more than odd ones, so of two slices of a tile one is parked well before the other arrives.
        jwg = xcd_order(bid - p.n_pers, p.sk_wgs);
        const int total = p.n_rem * nk;
        u = jwg * p.units_per_wg + ((jwg & 1) ? p.sk_skew : 0);
        u_end = jwg + 1 == p.sk_wgs ? total : (jwg + 1) * p.units_per_wg + ((jwg & 1) ? 0 : p.sk_skew);
        u = u < total ? u : total;
        u_end = u_end < total ? u_end : total;
    }

    // ---- fragment read offsets ----
    const int koff0 = FP8 ? ((2 * g) ^ (lane & 7)) << 4 : (g ^ (lane & 7)) << 4;     // bf16: k-step 0, k-step 1 = koff0 ^ 64; fp8: first / second half (koff0 ^ 16) of the fragment
    constexpr int KS1 = FP8 ? 16 : 64;
    const int a_rd = (wm * 64 + fr) * 128;                       // + mt*2048 within A_h
    const int b_rd = (wn * 32 + fr) * 128;                       // + nt*2048 within B_h
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int n_out_total = swiglu ? p.N / 2 : p.N;
    const bool vec_ok = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0);

    int seg_no = 0;
    VZ_STAMP(0)
    while (u < u_end) {
        int k0 = 0, k1 = nk;
        if (sk) {
            const int tr = u / nk;
            k0 = u - tr * nk;
            k1 = k0 + (u_end - u) < nk ? k0 + (u_end - u) : nk;
            tile = p.n_full + tr;
        }
        const int nks = k1 - k0;
        // tile index -> (bm, bn): row tiles in groups of 8, columns inside a group, rows fastest.  The 32 workgroups an XCD runs at
        // a time (consecutive indices) then cover 8 row panels x 4 column panels (12 operand panels through its L2 instead of
        // 33 when M is many tiles tall: the Q-Former K/V projection of 80 tiles re-read A 32 times).  tiles_m <= 8: plain
        // column-major, as before.
        const int per_group = 8 * p.tiles_n, grp = tile / per_group, in_grp = tile - grp * per_group;
        const int gsz = p.tiles_m - grp * 8 < 8 ? p.tiles_m - grp * 8 : 8;
        const int bn = in_grp / gsz, bm = grp * 8 + (in_grp - bn * gsz);

        // ---- staging sources: half-tile X_h, instruction j -> LDS chunk ch = j*512 + tid (row ch>>3, slot ch&7) ----
        const char* src[4][2];   // [A_0, A_1, B_0, B_1][j]
        auto tile_srcs = [&](int bm_, int bn_, int k0_, const char* (&out)[4][2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = j * 512 + tid;
                const int r = ch >> 3, c = ch & 7;
                const int gc = (c ^ (r & 7)) * 16 + k0_ * 128;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int arow = bm_ * 256 + (r >> 6) * 128 + h * 64 + (r & 63);     // LDS row r of A_h <-> wave row wm = r>>6
                    arow = arow < p.M ? arow : p.M - 1;
                    int wrow = bn_ * 256 + (r >> 5) * 64 + h * 32 + (r & 31);      // LDS row r of B_h <-> wave col wn = r>>5
                    wrow = wrow < p.N ? wrow : p.N - 1;
                    out[h][j] = (const char*)p.A + (size_t)arow * p.lda * ESZ + gc;
                    out[2 + h][j] = (const char*)p.W + (size_t)wrow * p.ldw * ESZ + gc;
                }
            }
        };
        tile_srcs(bm, bn, k0, src);
        // region: 0 A_0, 1 A_1, 2 B_0, 3 B_1
        auto stage_at = [&](int region, int kt, int slot_off) {
            const int t = kt < nks ? kt : nks - 1;
            char* dst = smem + slot_off + wave_off;
            const int kb = t * 128;
            glds16(src[region][0] + kb, dst);
            glds16(src[region][1] + kb, dst + 8192);
        };

        f32x4 acc[2][2][2][4];   // [qm][qn][nt][mt]
        VZ_ACC_FOR_EACH({ (void)i_; a_ = (f32x4){0.f, 0.f, 0.f, 0.f}; })

        // ---- prologue: K-tiles 0 and 1 issued, the alpha slots of tile 0 landed ----
        int rd = 0, st = 8 * HALF_BYTES;    // ring offsets of A_0(t) and of the next slot to stage (sequence 4t+8)
        stage_at(0, 0, 0); stage_at(2, 0, HALF_BYTES); stage_at(3, 0, 2 * HALF_BYTES); stage_at(1, 0, 3 * HALF_BYTES);
        stage_at(0, 1, 4 * HALF_BYTES); stage_at(2, 1, 5 * HALF_BYTES); stage_at(3, 1, 6 * HALF_BYTES); stage_at(1, 1, 7 * HALF_BYTES);
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // Stagger: the waves of the lower tile half (wm = 1, the second wave on every SIMD) run one barrier behind, so on
        // each SIMD one wave issues its ds_reads / DMAs while its partner owns the matrix pipe.  Every wave still executes
        // the same number of barriers (the other half takes the matching one after the loop); the hazard derivation in the
        // header holds with the reader one barrier later (its wait and its reads are two barriers apart).
        if (late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        __builtin_amdgcn_sched_barrier(0);
        if (seg_no < 3) { VZ_STAMP(1 + seg_no * 5) }
        if (seg_no == 0 && p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 14] = (long long)__builtin_amdgcn_s_memtime();   // core clock

        bf16x8 af[4][2], b0f[2][2], b1f[2][2];   // [mt][ks], [nt][ks]
        for (int t = 0; t < nks; ++t) {
            const char* pa0 = smem + rd;
            const char* pb0 = smem + ring_adv(rd, 1);
            const char* pb1 = smem + ring_adv(rd, 2);
            const char* pa1 = smem + ring_adv(rd, 3);
            // ================= phase alpha: quadrants (0,0) and (0,1) =================
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    b0f[nt][ks] = *(const bf16x8*)(pb0 + b_rd + nt * 2048 + (koff0 ^ (ks * KS1)));
                    b1f[nt][ks] = *(const bf16x8*)(pb1 + b_rd + nt * 2048 + (koff0 ^ (ks * KS1)));
                }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(pa0 + a_rd + mt * 2048 + (koff0 ^ (ks * KS1)));
            stage_at(0, t + 2, st);                  // A_0, B_0 of tile t+2 into the slots B_1, A_1 of tile t-1 left
            stage_at(2, t + 2, ring_adv(st, 1));
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            PHASE_SYNC_BEGIN()
            if constexpr (FP8) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[0][0][nt][mt] = mma2<true>(b0f[nt], af[mt], acc[0][0][nt][mt]);
                        acc[0][1][nt][mt] = mma2<true>(b1f[nt], af[mt], acc[0][1][nt][mt]);
                    }
            } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[0][0][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[nt][ks], af[mt][ks], acc[0][0][nt][mt], 0, 0, 0);
                        acc[0][1][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[nt][ks], af[mt][ks], acc[0][1][nt][mt], 0, 0, 0);
                    }
            }
            PHASE_SYNC_END()
            // ================= phase beta: quadrants (1,0) and (1,1), B fragments still in registers =================
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(pa1 + a_rd + mt * 2048 + (koff0 ^ (ks * KS1)));
            stage_at(3, t + 2, ring_adv(st, 2));     // B_1, A_1 of tile t+2 into the slots A_0, B_0 of this tile (read in alpha)
            stage_at(1, t + 2, ring_adv(st, 3));
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            rd = ring_adv(rd, 4);
            st = ring_adv(st, 4);
            PHASE_SYNC_BEGIN()
            if constexpr (FP8) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[1][0][nt][mt] = mma2<true>(b0f[nt], af[mt], acc[1][0][nt][mt]);
                        acc[1][1][nt][mt] = mma2<true>(b1f[nt], af[mt], acc[1][1][nt][mt]);
                    }
            } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        acc[1][0][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[nt][ks], af[mt][ks], acc[1][0][nt][mt], 0, 0, 0);
                        acc[1][1][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[nt][ks], af[mt][ks], acc[1][1][nt][mt], 0, 0, 0);
                    }
            }
            PHASE_SYNC_END()
        }
        if (!late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs must not outlive this use of the ring
        if (seg_no < 3) { VZ_STAMP(2 + seg_no * 5) }
        if (seg_no == 0 && p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 15] = (long long)__builtin_amdgcn_s_memtime();

        // ---- a K-slice of a tile: the last slice to arrive finishes the tile, the others park their partial sums ----
        bool finish = true;
        if (nks != nk) {
            const int tr = tile - p.n_full;
            const int U = p.units_per_wg;
            int j_first = (tr * nk) / U, j_last = (tr * nk + nk - 1) / U;          // owners of the tile's first / last K-tile
            if ((j_first & 1) && tr * nk < j_first * U + p.sk_skew) --j_first;
            if ((j_last & 1) && tr * nk + nk - 1 < j_last * U + p.sk_skew) --j_last;
            const int nseg = j_last - j_first + 1, seg = jwg - j_first;
            // slot of (workgroup j, tile tr) = j + tr: unique (along the unit axis either j or tr steps), and the slices of
            // one tile are consecutive; at most sk_wgs + n_rem slots
            float* slots = p.ws + (size_t)(j_first + tr) * TILE_FLOATS;
            float* mine = slots + (size_t)seg * TILE_FLOATS;
            // Arrival ticket first: it only decides who finishes the tile (the slice that arrives last).  Every other slice
            // is parked with write-through (sc1) stores, drained by every wave before ONE lane bumps the tile's `ready`
            // counter; the finisher reads with sc1 loads once `ready` shows all of them.  It only ever waits for workgroups
            // that have already taken their arrival ticket, i.e. that are past their main loop, resident, and need nothing
            // from anyone to complete their stores: the wait is bounded by one 256 KiB write.  No release/acquire fences:
            // on gfx950 those write back / invalidate the XCD's L2, which the other workgroups' operand streams live in
            // (measured: 2.6x slower GEMM with __threadfence()).
            unsigned* arrive = (unsigned*)p.tickets + 2 * tr;
            unsigned* ready = arrive + 1;
            int* flag = (int*)smem;     // the ring is idle here
            if (tid == 0) {
                const unsigned before = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = before == (unsigned)(nseg - 1);
                // more arrivals than the tile has slices: the pair was not zero when this launch began (see the finisher's wait)
                if (before >= (unsigned)nseg && p.err) __hip_atomic_store(p.err, VZ_ASYNC_STREAMK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            finish = __builtin_amdgcn_readfirstlane(*flag) != 0;
            if (!finish || nseg > 2) {       // (with more than two slices the finisher parks its own too: fixed summation order)
                VZ_ACC_FOR_EACH({ st2_sc1(mine + (size_t)((i_ * 2) * 512 + tid) * 2, a_[0], a_[1]);
                                  st2_sc1(mine + (size_t)((i_ * 2 + 1) * 512 + tid) * 2, a_[2], a_[3]); })
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && !finish) __hip_atomic_fetch_add(ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (finish) {
                if (tid == 0) {
                    // bounded (guide section 5.6 "bound every spin"): the slices waited for are already past their main loop, so the
                    // wait is one 256 KiB write.  A count that never arrives means the {arrive, ready} pair of this tile was not what
                    // this launch assumed (another launch on the same workspace at the same time, or counters left by a launch that
                    // was torn down): the finisher then raises the async error word, leaves BOTH counters as they are (a late slice
                    // must not find a freshly zeroed `ready` to bump - that would hand the next launch a count for slots nobody has
                    // written) and poisons the tile with NaN instead of summing stale slots.  The host resets the counters when it
                    // reads the error (vz_engine_async_error / vz_op_async_error).
                    int spins = 0;
                    while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(nseg - 1) && spins < (1 << 22)) {
                        __builtin_amdgcn_s_sleep(4);
                        ++spins;
                    }
                    const bool ok = spins < (1 << 22);
                    if (ok) {
                        __hip_atomic_store(ready, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // both counters ready for
                        __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the next launch
                    } else if (p.err) {
                        __hip_atomic_store(p.err, VZ_ASYNC_STREAMK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    flag[1] = ok ? 1 : 0;
                }
                __syncthreads();        // everyone loads behind the lane that saw the count
                if (__builtin_amdgcn_readfirstlane(flag[1]) == 0) {
                    VZ_ACC_FOR_EACH({ (void)i_; a_ = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")}; })
                } else if (nseg == 2) {
                    const float* other = slots + (size_t)(seg ^ 1) * TILE_FLOATS;
                    VZ_ACC_ADD_FROM(other)
                } else {        // fixed slice order: the sum does not depend on who arrived last
                    VZ_ACC_FOR_EACH({ (void)i_; a_ = (f32x4){0.f, 0.f, 0.f, 0.f}; })
                    for (int s2 = 0; s2 < nseg; ++s2) {
                        const float* other = slots + (size_t)s2 * TILE_FLOATS;
                        VZ_ACC_ADD_FROM(other)
                    }
                }
            }
            __syncthreads();            // the flag word is ring space again
            if (seg_no < 3) { VZ_STAMP(3 + seg_no * 5) }
        }

        // ---- epilogue: acc[qm][qn][nt][mt][j] = C[m][n], m = .. + qm*64 + mt*16 + fr, n = .. + qn*32 + nt*16 + 4g + j ----
        if constexpr (FP8) {
            if (finish) {       // the two row scales (powers of two: exact) once, on the complete fp32 sums
                const int m0 = bm * 256 + wm * 128 + fr, n0 = bn * 256 + wn * 64 + 4 * g;
                float sx[2][4];
#pragma unroll
                for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) { const int m = m0 + qm * 64 + mt * 16; sx[qm][mt] = p.ascale[m < p.M ? m : p.M - 1]; }
#pragma unroll
                for (int qn = 0; qn < 2; ++qn)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        float sw[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const int n = n0 + qn * 32 + nt * 16 + j; sw[j] = p.wscale[n < p.N ? n : p.N - 1]; }
#pragma unroll
                        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                                for (int j = 0; j < 4; ++j) acc[qm][qn][nt][mt][j] *= sx[qm][mt] * sw[j];
                    }
            }
        }
        if (finish) {
            const int m_base = bm * 256 + wm * 128 + fr, n_base = bn * 256 + wn * 64;
            const bool interior = bn * 256 + 256 <= p.N && vec_ok && (((uintptr_t)p.bias) & 15) == 0;
            if (interior) {
                switch (p.act) {
                    case VZ_ACT_QUICK_GELU: epilogue_fast<VZ_ACT_QUICK_GELU>(p, acc, m_base, n_base, g); break;
                    case VZ_ACT_GELU_ERF: epilogue_fast<VZ_ACT_GELU_ERF>(p, acc, m_base, n_base, g); break;
                    case VZ_ACT_SWIGLU: epilogue_fast<VZ_ACT_SWIGLU>(p, acc, m_base, n_base, g); break;
                    default: epilogue_fast<VZ_ACT_NONE>(p, acc, m_base, n_base, g); break;
                }
            } else {
            // explicit instances: hipcc does not unroll a (qm, mt) loop around this body, and a rolled loop would index the
            // accumulators dynamically (= a scratch copy of all 128 registers)
            epilogue_rows<0, 0>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<0, 1>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<0, 2>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<0, 3>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<1, 0>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<1, 1>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<1, 2>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            epilogue_rows<1, 3>(p, acc, m_base, n_base, g, swiglu, n_out_total, vec_ok);
            }
        }
        if ((p.stamps || p.drain) && seg_no < 3) {       // profiling only: the stamp is taken once the stores have drained
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            VZ_STAMP(4 + seg_no * 5)
            if (p.stamps && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 5 + seg_no * 5] = ((long long)nks << 32) | (finish ? 1 : 0) | (nks != nk ? 2 : 0);
        }
        ++seg_no;
        u += nks;
        if (!sk && u >= u_end && t_lin + p.n_pers < p.n_full) {     // persistent: this workgroup's next whole tile
            t_lin += p.n_pers;
            tile = xcd_order(t_lin, p.n_full);
            u = 0;
        }
        if (u < u_end) __syncthreads();     // next slice / next tile re-stages the ring from slot 0
    }
}

__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(Gemm256Params p) { gemm256_body<false>(p); }
__global__ __launch_bounds__(512, 2) void gemm256_fp8_kernel(Gemm256Params p) { gemm256_body<true>(p); }

long long* g_stamps;         // 16 stamps per workgroup of the last launch (profiling knob 6)
int g_stamp_wgs;
int g_num_cu;

// Stream-K state = {fp32 slots, {arrive, ready} per remainder tile, async error word}.  Launches that share it must be ordered,
// so there is one per (device, stream): two engines, two streams or two devices in one process never meet on one set of
// tickets (round 1 had ONE process-wide set - see DESIGN.md section 5b).  Allocated on first use of a stream by a launch that
// takes the stream-K path, never inside a stream capture (the capturing launcher gets whole tiles only).
struct SkState { float* ws = nullptr; int* tickets = nullptr; int* err = nullptr; size_t ws_bytes = 0; int n_tickets = 0; };
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, SkState> g_sk;

int sk_state_for(hipStream_t s, SkState** out) {
    int dev = 0;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_sk_mu);
    SkState& st = g_sk[{dev, s}];
    if (!st.ws) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs != hipStreamCaptureStatusNone) { *out = nullptr; return VZ_OK; }       // no allocation inside a capture
        st.ws_bytes = (size_t)2 * g_num_cu * TILE_FLOATS * sizeof(float);               // <= P + n_rem slots are ever in use
        st.n_tickets = 2 * g_num_cu;                                                    // {arrive, ready} per remainder tile
        VZ_CHECK_HIP(hipMalloc((void**)&st.ws, st.ws_bytes));
        VZ_CHECK_HIP(hipMalloc((void**)&st.tickets, (size_t)(st.n_tickets + 4) * sizeof(int)));
        VZ_CHECK_HIP(hipMemset(st.tickets, 0, (size_t)(st.n_tickets + 4) * sizeof(int)));
        st.err = st.tickets + st.n_tickets;
    }
    *out = &st;
    return VZ_OK;
}

}  // namespace

int g_gemm256_streamk = 1;   // vz_tune_set(4, v): 1 = stream-K tail (default), 0 = whole tiles only
int g_gemm256_stamps = 0;    // vz_tune_set(6, v): 1 = record in-kernel phase stamps (vz_prof_gemm_stamps)
int g_gemm256_drain = 0;     // vz_tune_set(11, v)
int g_gemm256_persist = 1;   // vz_tune_set(34, v)
int g_gemm256_skew = 2;      // vz_tune_set(5, v): K-tiles by which even / odd stream-K workgroups lead / lag

int vz_init_gemm256_kernel() {
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES));
    int dev = 0;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    VZ_CHECK_HIP(hipDeviceGetAttribute(&g_num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    VZ_CHECK_HIP(hipMalloc((void**)&g_stamps, (size_t)4096 * 16 * sizeof(long long)));
    VZ_CHECK_HIP(hipDeviceSynchronize());
    return VZ_OK;
}

// tile plan + stream-K tail for a filled parameter block (nk = K-tiles of this element type), then the launch
static int plan_and_launch(Gemm256Params& p, int nk, int* err_word, bool fp8, hipStream_t s) {
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    const int T = p.tiles_m * p.tiles_n, P = g_num_cu;
    p.n_full = T; p.n_rem = 0; p.sk_wgs = 0; p.units_per_wg = 1; p.sk_skew = 0;
    p.ws = nullptr; p.tickets = nullptr; p.err = nullptr;
    const int rem = T % P;
    // a last round that fills at most half of the CUs is cut along K instead.  Measured (S=2048): down-proj (128 tiles,
    // K=14336) 250 -> 190 us, Q-Former cross-attention K/V (384 tiles) 227 -> 214 us; a fuller tail (QKV, 192 tiles) does
    // not pay for the 256 KiB-per-tile fix-up (100 -> 117 us).  g_gemm256_streamk = 2 forces it for any tail (tests).
    if (g_gemm256_streamk && rem > 0 && (rem * 2 <= P || g_gemm256_streamk == 2)) {
        const long units = (long)rem * nk;
        long wgs = units / SK_MIN_UNITS;
        if (wgs > P) wgs = P;
        if (wgs >= 1) {
            const int U = (int)((units + wgs - 1) / wgs);
            wgs = (units + U - 1) / U;
            SkState* st = nullptr;
            { int r = sk_state_for(s, &st); if (r) return r; }
            if (st && (size_t)(wgs + rem) * TILE_FLOATS * sizeof(float) <= st->ws_bytes && 2 * rem <= st->n_tickets) {
                p.n_full = T - rem; p.n_rem = rem; p.sk_wgs = (int)wgs; p.units_per_wg = U;
                p.sk_skew = U >= 24 ? g_gemm256_skew : 0;
                p.ws = st->ws; p.tickets = st->tickets; p.err = err_word ? err_word : st->err;
            }
        }
    }
    // persistent whole tiles (round 3; vz_tune_set(34, 0) = one workgroup per tile again): with more whole tiles than CUs, P workgroups walk them
    // (tile b, b + P, ...) instead of n_full workgroups taking one each - no workgroup start / LDS allocation / teardown per tile.  Measured in one
    // process (tools/bench_kernels.py gemmsq): Stage-1 gate|up (12736 rows, 5600 tiles) 2344 -> 2240 us, Stage-1 QKV 509 -> 496, CLIP fc1 at 64 x 577 rows
    // 392 -> 376, 4 rounds of Zephyr tiles 446 -> 432; one-round shapes unchanged.  Also built: requesting the NEXT tile's first two K-tiles before the
    // current tile's epilogue (their HBM latency under the store drain) - no gain over plain persistence (2261 / 503 / 382 / 446 us): removed.
    p.n_pers = (g_gemm256_persist && p.n_full > P) ? P : p.n_full;
    p.stamps = nullptr; p.drain = g_gemm256_drain;
    if (g_gemm256_stamps && p.n_pers + p.sk_wgs <= 4096) {
        p.stamps = g_stamps; g_stamp_wgs = p.n_pers + p.sk_wgs;
        VZ_CHECK_HIP(hipMemsetAsync(g_stamps, 0, (size_t)g_stamp_wgs * 16 * sizeof(long long), s));
    }
    if (fp8) vz_launch_timed(gemm256_fp8_kernel, dim3(p.n_pers + p.sk_wgs), dim3(512), RING_BYTES, s, p);
    else vz_launch_timed(gemm256_bf16_kernel, dim3(p.n_pers + p.sk_wgs), dim3(512), RING_BYTES, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_gemm256(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(!a.norm_w, "linear: fused RMSNorm prologue exists on the GEMV path only");
    { int r = vz_init_gemm256_kernel(); if (r) return r; }
    Gemm256Params p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32; p.ascale = nullptr; p.wscale = nullptr;
    return plan_and_launch(p, a.K >> 6, a.err, false, s);
}

// e4m3 x e4m3 on the same 8-phase pipeline (gemm_fp8.hip dispatches here for grids that fill the chip): K % 128 == 0
int vz_launch_gemm256_fp8(const Fp8LinearArgs& a, hipStream_t s) {
    { int r = vz_init_gemm256_kernel(); if (r) return r; }
    Gemm256Params p;
    p.A = (const bf16_t*)a.A8; p.W = (const bf16_t*)a.W8; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32; p.ascale = a.ascale; p.wscale = a.wscale;
    return plan_and_launch(p, a.K >> 7, nullptr, true, s);
}

// profiling: copy the phase stamps of the last stamped 256^2 launch (16 int64 per workgroup) to the host
int vz_gemm256_read_stamps(long long* host, int max_wgs, int* n_wgs) {
    VZ_CHECK_ARG(host && n_wgs, "stamps: null argument");
    VZ_CHECK_HIP(hipDeviceSynchronize());
    const int n = g_stamp_wgs < max_wgs ? g_stamp_wgs : max_wgs;
    if (n > 0) VZ_CHECK_HIP(hipMemcpy(host, g_stamps, (size_t)n * 16 * sizeof(long long), hipMemcpyDeviceToHost));
    *n_wgs = n;
    return VZ_OK;
}

// Async error of the stream-K fix-up on `s` (op-level launches; an engine passes its own word through LinearArgs.err and reports
// it through vz_engine_async_error).  Blocking: waits for the stream, reads the word, and when it is set clears it AND the
// tickets of that stream so the next launch starts from clean counters.
int vz_gemm256_async_error(hipStream_t s, int* err, bool reset_only) {
    *err = 0;
    int dev = 0;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    SkState st;
    {
        std::lock_guard<std::mutex> lk(g_sk_mu);
        auto it = g_sk.find({dev, s});
        if (it == g_sk.end() || !it->second.ws) return VZ_OK;
        st = it->second;
    }
    VZ_CHECK_HIP(hipStreamSynchronize(s));
    if (!reset_only) VZ_CHECK_HIP(hipMemcpy(err, st.err, sizeof(int), hipMemcpyDeviceToHost));
    if (*err || reset_only) VZ_CHECK_HIP(hipMemset(st.tickets, 0, (size_t)(st.n_tickets + 4) * sizeof(int)));
    return VZ_OK;
}

// test hook: leave the {arrive, ready} pair of remainder tile `tr` on stream `s` in a state no launch can complete from
int vz_gemm256_corrupt_tickets(hipStream_t s, int tr, int arrive, int ready) {
    int dev = 0;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    SkState* st = nullptr;
    { int r = sk_state_for(s, &st); if (r) return r; }
    VZ_CHECK_ARG(st && tr >= 0 && 2 * tr + 1 < st->n_tickets, "corrupt_tickets: bad tile");
    const int v[2] = {arrive, ready};
    VZ_CHECK_HIP(hipStreamSynchronize(s));
    VZ_CHECK_HIP(hipMemcpy(st->tickets + 2 * tr, v, sizeof(v), hipMemcpyHostToDevice));
    return VZ_OK;
}
