// 256x256x64 deep-pipelined bf16 MFMA GEMM for gfx950:  C[M,N'] = epi(A[M,K] . W[N,K]^T)
//
// The 128x128 kernel (gemm.hip) needs the full LDS read bandwidth of a CU to feed its MFMAs (64x64 wave
// tiles: 512 B of LDS per 16x16x32 MFMA) and stalls on one vmcnt(0)+barrier per K-tile: ~36-39 % of
// the 2.5 PF peak.  This kernel follows the 8-phase structure of the CDNA4 guide (section 5, "The 256^2
// 8-phase template"):
//   * 8 waves (2 M x 4 N), each owning a 128x64 output tile = 32 accumulators of 16x16 (384 B of LDS
//     per MFMA), one workgroup per CU, 128 KiB of LDS = 2 K-tile buffers x {A_0, A_1, B_0, B_1};
//     a half-tile X_h holds the rows every wave needs for its output quadrant h (128 rows x 64 k).
//   * each K-tile is 2 phases of 32 MFMAs (two output quadrants each); measured on MI355X the 4-phase form of the
//     guide spends more time in barrier/LDS latency than in MFMAs with this staging, the 2-phase form is faster:
//        phase alpha: read A_0 (8x ds_read_b128), B_0, B_1 (4x each)  MFMA quadrants (0,0),(0,1)  stage B_1(t+1), A_1(t+1)
//        phase beta : read A_1 (8x), B fragments stay in registers    MFMA quadrants (1,0),(1,1)  stage A_0(t+2), B_0(t+2), vmcnt(4)
//     every phase = {ds_reads, 4 LDS-DMA per thread, lgkmcnt(0), s_barrier, 32 MFMA, s_barrier}; the two wave groups
//     (upper / lower half of the tile) run one barrier apart, so one group's loads overlap the other's MFMAs.
//   * operands are staged with 16-byte LDS-DMA (global_load_lds_dwordx4) that stays in flight ACROSS the raw
//     s_barriers: the only VMEM wait in the loop is one counted `s_waitcnt vmcnt(4)` per K-tile, which
//     leaves the two newest half-tiles in flight.  Hazards:
//       RAW  every half-tile of K-tile t+1 is issued no later than phase alpha of tile t, retired by the vmcnt(4) of
//            phase beta (only the 4 DMAs of that phase are newer), and first read in phase alpha of tile t+1, i.e.
//            two barriers after the wait (one more than the stagger needs);
//       WAR  a region is re-staged at the earliest one phase after its last ds_read, and every wave has
//            passed its lgkmcnt(0) and the phase-end barrier by then.
//     K-tiles past the end are clamped to the last one (identical bytes re-written), so the loop has no
//     tail variants and vmcnt(4) is exact in every iteration.
//   * LDS image lane-linear per DMA instruction; XOR swizzle (chunk ^= row & 7) on the SOURCE address and on
//     the ds_read_b128; XCD-aware bijective tile order; same fused epilogues as gemm.hip.
#include "vz_common.h"

namespace {

constexpr int HALF_BYTES = 128 * 64 * 2;       // 16 KiB: 128 rows x 64 k
constexpr int BUF_BYTES = 4 * HALF_BYTES;      // A_0 A_1 B_0 B_1
constexpr int LDS_BYTES = 2 * BUF_BYTES;       // 128 KiB

struct Gemm256Params {
    const bf16_t* A; const bf16_t* W; void* C;
    const float* bias; const bf16_t* residual;
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32, tiles_m, tiles_n;
};

__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
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

// A phase = {ds_reads, LDS-DMA issue, [vmcnt], lgkmcnt(0), s_barrier | 16 MFMA, s_barrier}.  The lgkmcnt(0) sits BEFORE the
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

__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(Gemm256Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;      // 2 x 4 waves, 128(m) x 64(n) each
    const int fr = lane & 15, g = lane >> 4;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r8 = nwg & 7;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
    const int bn = tile / p.tiles_m, bm = tile - bn * p.tiles_m;

    // ---- staging sources: half-tile X_h, instruction j -> LDS chunk ch = j*512 + tid (row ch>>3, slot ch&7) ----
    const char* src[4][2];   // [A_0, A_1, B_0, B_1][j]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = j * 512 + tid;
        const int r = ch >> 3, c = ch & 7;
        const int gc = (c ^ (r & 7)) * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int arow = bm * 256 + (r >> 6) * 128 + h * 64 + (r & 63);     // LDS row r of A_h <-> wave row wm = r>>6
            arow = arow < p.M ? arow : p.M - 1;
            int wrow = bn * 256 + (r >> 5) * 64 + h * 32 + (r & 31);      // LDS row r of B_h <-> wave col wn = r>>5
            wrow = wrow < p.N ? wrow : p.N - 1;
            src[h][j] = (const char*)p.A + (size_t)arow * p.lda * 2 + gc;
            src[2 + h][j] = (const char*)p.W + (size_t)wrow * p.ldw * 2 + gc;
        }
    }
    const int nk = p.K >> 6;
    const int wave_off = wave * 1024;
    // region: 0 A_0, 1 A_1, 2 B_0, 3 B_1
    auto stage = [&](int region, int kt) {
        const int t = kt < nk ? kt : nk - 1;
        char* dst = smem + (t & 1) * BUF_BYTES + region * HALF_BYTES + wave_off;
        const int kb = t * 128;
        glds16(src[region][0] + kb, dst);
        glds16(src[region][1] + kb, dst + 8192);
    };

    // ---- fragment read offsets ----
    const int koff0 = (g ^ (lane & 7)) << 4;                     // k-step 0; k-step 1 = koff0 ^ 64
    const int a_rd = (wm * 64 + fr) * 128;                       // + mt*2048 within A_h
    const int b_rd = (wn * 32 + fr) * 128;                       // + nt*2048 within B_h

    f32x4 acc[2][2][2][4];   // [qm][qn][nt][mt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[a][b][c][d] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K-tile 0 complete, K-tile 1 minus A_1 ----
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
    stage(0, 1); stage(2, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // Stagger: the waves of the lower tile half (wm = 1, the second wave on every SIMD) run one barrier behind, so on
    // each SIMD one wave issues its ds_reads / DMAs while its partner owns the matrix pipe.  Every wave still executes
    // the same number of barriers (the other half takes the matching one after the loop); the hazard derivation in the
    // header holds with the reader one barrier later (its wait and its reads are two barriers apart).
    const bool late = __builtin_amdgcn_readfirstlane(wm) != 0;
    if (late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    __builtin_amdgcn_sched_barrier(0);

    bf16x8 af[4][2], b0f[2][2], b1f[2][2];   // [mt][ks], [nt][ks]
    for (int t = 0; t < nk; ++t) {
        const char* base = smem + (t & 1) * BUF_BYTES;
        // ================= phase alpha: quadrants (0,0) and (0,1) =================
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                b0f[nt][ks] = *(const bf16x8*)(base + 2 * HALF_BYTES + b_rd + nt * 2048 + (koff0 ^ (ks * 64)));
                b1f[nt][ks] = *(const bf16x8*)(base + 3 * HALF_BYTES + b_rd + nt * 2048 + (koff0 ^ (ks * 64)));
            }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(base + a_rd + mt * 2048 + (koff0 ^ (ks * 64)));
        stage(3, t + 1);     // B_1, A_1 of the NEXT tile go into the other buffer (last read two / one phase ago)
        stage(1, t + 1);
        PHASE_SYNC_BEGIN()
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[0][0][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[nt][ks], af[mt][ks], acc[0][0][nt][mt], 0, 0, 0);
                    acc[0][1][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[nt][ks], af[mt][ks], acc[0][1][nt][mt], 0, 0, 0);
                }
        PHASE_SYNC_END()
        // ================= phase beta: quadrants (1,0) and (1,1), B fragments still in registers =================
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(base + HALF_BYTES + a_rd + mt * 2048 + (koff0 ^ (ks * 64)));
        stage(0, t + 2);     // A_0, B_0 of tile t+2 overwrite this tile's copies (read in phase alpha, retired before its barrier)
        stage(2, t + 2);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but the 4 DMAs just issued have landed: K-tile t+1 is complete
        PHASE_SYNC_BEGIN()
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[1][0][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0f[nt][ks], af[mt][ks], acc[1][0][nt][mt], 0, 0, 0);
                    acc[1][1][nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1f[nt][ks], af[mt][ks], acc[1][1][nt][mt], 0, 0, 0);
                }
        PHASE_SYNC_END()
    }
    if (!late) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail DMAs must not outlive the workgroup's LDS

    // ---- epilogue: acc[qm][qn][nt][mt][j] = C[m][n], m = .. + qm*64 + mt*16 + fr, n = .. + qn*32 + nt*16 + 4g + j ----
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int n_out_total = swiglu ? p.N / 2 : p.N;
    const bool vec_ok = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0);
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = bm * 256 + wm * 128 + qm * 64 + mt * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                const int nb = bn * 256 + wn * 64 + qn * 32;
                if (swiglu) {   // nt 0 = 16 gate rows, nt 1 = the matching 16 up rows
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_silu(acc[qm][qn][0][mt][j]) * acc[qm][qn][1][mt][j];
                    store4(p, m, (nb >> 1) + g * 4, v, n_out_total, vec_ok);
                } else {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int n0 = nb + nt * 16 + g * 4;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float tv = acc[qm][qn][nt][mt][j];
                            if (p.bias && n0 + j < p.N) tv += p.bias[n0 + j];
                            v[j] = apply_act(tv, p.act);
                        }
                        store4(p, m, n0, v, n_out_total, vec_ok);
                    }
                }
            }
        }
}

}  // namespace

int vz_init_gemm256_kernel() {
    static bool done = false;
    if (done) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemm256_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    done = true;
    return VZ_OK;
}

int vz_launch_gemm256(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(!a.norm_w, "linear: fused RMSNorm prologue exists on the GEMV path only");
    Gemm256Params p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32;
    p.tiles_m = (a.M + 255) / 256;
    p.tiles_n = (a.N + 255) / 256;
    { int r = vz_init_gemm256_kernel(); if (r) return r; }
    vz_launch_timed(gemm256_bf16_kernel, dim3(p.tiles_m * p.tiles_n), dim3(512), LDS_BYTES, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
