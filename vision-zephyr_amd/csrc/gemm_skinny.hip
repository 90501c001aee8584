// Weight-streaming MFMA GEMM for 2..16 activation rows (batched decode): C[M<=16, N'] = epi(x[M,K] . W[N,K]^T).
//
// The GEMV (gemv.hip) multiplies every weight by every activation row on the VALU: v_dot2c work per weight byte grows
// with M and a 4-row launch already takes 1.4x a 1-row launch.  Here the 16 x 16 x 32 MFMA takes a 16-row x 32-k weight
// tile as its A operand and ALL activation rows (padded to 16) as its B operand: one matrix instruction per KiB of
// weights whatever M is, no cross-lane reduction (lane (m, g) of the result holds outputs n = 4g..4g+3 of row m), and the
// kernel stays a pure HBM stream up to 16 rows.
//
//   * workgroup = 8 waves = one group of 16 weight rows (SwiGLU: the 16 gate + 16 up rows of one interleaved group) over
//     the whole K; wave w streams the contiguous K-slice w, so the 256 row groups of a 4096-row projection still put 2048
//     waves on the chip; the eight partial tiles meet in LDS (8 KiB) and wave 0 runs the epilogue.
//   * a step is 64 k = two MFMAs: lane (row r, g) loads the 32 contiguous bytes W[r][64s + 16g .. 64s + 16g + 15] - four
//     lanes take one whole 128-byte line, every line is requested once - and feeds k = 16g..16g+7 to the first MFMA,
//     16g+8..16g+15 to the second; the activation fragments use the same k assignment, which is all the contraction needs.
//   * activations: with a fused RMSNorm (QKV, gate-up, lm_head: K = hidden) the normalised rows are staged once per
//     workgroup in LDS (one wave per row, row stride K + 8 elements: conflict-free fragment reads); without it (O-proj,
//     down-proj) the B fragments are loaded straight from global memory - x is at most 16 x 14336 bf16 = 448 KiB, L2
//     resident, and rides in the same prefetch ring as the weights.
//   * FP8 weights (W8A16, vz_hip/quant.py): a lane's 16 bytes are the same 16 k of the step; they are widened to the two bf16
//     fragments in registers (exact) and the row's power-of-two scale multiplies the fp32 result once.
#include "vz_common.h"

int g_skinny_even = 1;          // vz_tune_set(35, 0): one persistent workgroup per CU whatever the group count (A/B)
namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// NW = waves per workgroup = K-slices of one row group: 8 while the projection has fewer than 512 row groups (4096 rows = 256
// groups -> 2048 waves), 4 beyond (gate-up: 896 groups fit the chip in ONE round of 4-wave workgroups instead of 1.75 rounds of
// 8-wave ones)
// 64-k steps in flight per wave (32 B per lane and operand each): sized so that every variant stays under 128 VGPRs = two
// 8-wave workgroups per CU
template <bool SWIGLU, bool NORM> struct Depth { static constexpr int U = (SWIGLU || !NORM) ? 4 : 8; };

struct SkinnyParams {
    const bf16_t* A; const bf16_t* W; void* C;
    const float* bias; const bf16_t* residual; const float* norm_w;
    const unsigned char* W8; const float* wscale;     // FP8 instantiation: e4m3 rows [N][ldw bytes] + fp32 2^e per row (vz_hip/quant.py)
    const bf16_t* Wt;                                 // bf16 weights in fragment order (tile_weights_kernel), or null: row-major W
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32;
    float norm_eps;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// 16 e4m3 weights (k = 16g .. 16g+15 of one row) -> the two bf16 A fragments of a 64-k step; exact (v_cvt_scalef32_pk_bf16_fp8)
__device__ __forceinline__ unsigned fp8x2_bf16x2(unsigned w, bool hi_half) {
    return hi_half ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                   : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
}
__device__ __forceinline__ void widen_fp8(const u32x4 w, u32x4& lo, u32x4& hi) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lo = (u32x4){fp8x2_bf16x2(w0, false), fp8x2_bf16x2(w0, true), fp8x2_bf16x2(w1, false), fp8x2_bf16x2(w1, true)};
    hi = (u32x4){fp8x2_bf16x2(w2, false), fp8x2_bf16x2(w2, true), fp8x2_bf16x2(w3, false), fp8x2_bf16x2(w3, true)};
}

// Fragment-tiled weights: the 16 bytes lane (r = lane & 15, g = lane >> 4) feeds to MFMA j (0, 1) of 64-k step s of row group G -
// W[16 G + r][64 s + 16 g + 8 j .. + 7] - sit at Wt[(((G * S + s) * 2 + j) * 64 + lane) * 8], S = K / 64: every wave-instruction of the
// weight stream reads 1 KiB contiguous (row-major: 16 rows x 64 bytes), values and k order unchanged - results are bit-identical.
__global__ __launch_bounds__(256) void tile_weights_kernel(const bf16_t* __restrict__ W, int ldw, bf16_t* __restrict__ Wt, int N, int K) {
    const int S = K >> 6;
    const long total = (long)(N >> 4) * S * 128;          // 16-byte chunks
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63), j = (int)((i >> 6) & 1);
        const long gs = i >> 7;
        const int s = (int)(gs % S), G = (int)(gs / S);
        const int r = lane & 15, g = lane >> 4;
        *(uint4*)(Wt + i * 8) = *(const uint4*)(W + (size_t)(16 * G + r) * ldw + 64 * s + 16 * g + 8 * j);
    }
}

// per-lane base, step stride and second-half offset (elements) of the weight stream in either layout
struct WAddr { const bf16_t* a; const bf16_t* b; int sstr, hoff; };
template <bool SWIGLU>
__device__ __forceinline__ WAddr waddr(const SkinnyParams& p, int grp, int lane) {
    const int fr = lane & 15, g = lane >> 4;
    WAddr w;
    if (p.Wt) {
        const size_t per_group = (size_t)(p.K >> 6) * 1024;
        w.a = p.Wt + (size_t)(SWIGLU ? 2 * grp : grp) * per_group + lane * 8;
        w.b = w.a + per_group;
        w.sstr = 1024; w.hoff = 512;
    } else {
        int row_a, row_b = 0;
        if (SWIGLU) { row_a = grp * 32 + fr; row_b = row_a + 16; }
        else { row_a = grp * 16 + fr; row_a = row_a < p.N ? row_a : p.N - 1; }
        w.a = p.W + (size_t)row_a * p.ldw + g * 16;
        w.b = p.W + (size_t)row_b * p.ldw + g * 16;
        w.sstr = 64; w.hoff = 8;
    }
    return w;
}

template <bool SWIGLU, bool NORM, int NW, bool FP8 = false>
__global__ __launch_bounds__(NW * 64) void skinny_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int K = p.K;
    const int xs_stride = K + 8;                                          // elements; 16 extra bytes per row: no bank conflicts
    bf16_t* xs = (bf16_t*)smem;                                           // NORM: [M][K + 8]
    float* red = (float*)(smem + (NORM ? (size_t)p.M * xs_stride * 2 : 0));  // [NW][2][64][4]

    // ---- weight rows of this workgroup ----
    const int grp = blockIdx.x;
    int row_a, row_b = 0;                                                  // row this lane streams (tile a; SwiGLU: tile b = up)
    if (SWIGLU) { row_a = grp * 32 + fr; row_b = row_a + 16; }
    else { row_a = grp * 16 + fr; row_a = row_a < p.N ? row_a : p.N - 1; }
    const WAddr wad = waddr<SWIGLU>(p, grp, lane);
    const bf16_t* wa = wad.a;
    const bf16_t* wb = wad.b;
    const int sstr = wad.sstr, hoff = wad.hoff;
    const unsigned char* wa8 = p.W8 + (size_t)row_a * p.ldw + g * 16;      // FP8: 16 bytes = the same 16 k of a step
    const unsigned char* wb8 = p.W8 + (size_t)row_b * p.ldw + g * 16;

    // ---- this wave's K-slice, in steps of 64 k ----
    constexpr int U = Depth<SWIGLU, NORM>::U;
    const int steps = K >> 6;
    const int per = (steps + NW - 1) / NW;
    const int s0 = wave * per < steps ? wave * per : steps;
    const int s1 = s0 + per < steps ? s0 + per : steps;

    u32x4 qa[U][2], qb[U][2], qx[U][2];
    const bool row_ok = fr < p.M;
    const bf16_t* xg = p.A + (size_t)(row_ok ? fr : 0) * p.lda + g * 16;   // !NORM: B fragments straight from global / L2
    auto issue = [&](int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = i0 + u;
            if (s < s1) {
                if (FP8) {
                    qa[u][0] = *(const u32x4*)(wa8 + (size_t)s * 64);
                    if (SWIGLU) qb[u][0] = *(const u32x4*)(wb8 + (size_t)s * 64);
                } else {
                    const bf16_t* pa = wa + (size_t)s * sstr;
                    qa[u][0] = *(const u32x4*)pa; qa[u][1] = *(const u32x4*)(pa + hoff);        // default cache policy (tiled copy: non-temporal measured equal)
                    if (SWIGLU) {
                        const bf16_t* pb = wb + (size_t)s * sstr;
                        qb[u][0] = *(const u32x4*)pb; qb[u][1] = *(const u32x4*)(pb + hoff);
                    }
                }
                if (!NORM) {
                    const u32x4* px = (const u32x4*)(xg + (size_t)s * 64);
                    qx[u][0] = row_ok ? px[0] : (u32x4){0u, 0u, 0u, 0u};
                    qx[u][1] = row_ok ? px[1] : (u32x4){0u, 0u, 0u, 0u};
                }
            }
        }
    };
    issue(s0);          // the first weights are in flight before the norm prologue

    if (NORM) {         // one wave per activation row (rows m = wave, wave + NW, ...): bf16(norm_w * x * rstd) -> LDS
        for (int m = wave; m < p.M; m += NW) {
            const bf16_t* x = p.A + (size_t)m * p.lda;
            float ss = 0.f;
            for (int k = lane * 8; k < K; k += 64 * 8) {
                const u16x8 v = *(const u16x8*)(x + k);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(v[j]); ss += f * f; }
            }
            const float rstd = rsqrtf(wave_sum(ss) / (float)K + p.norm_eps);
            for (int k = lane * 8; k < K; k += 64 * 8) {
                const u16x8 v = *(const u16x8*)(x + k);
                const f32x4 w0 = *(const f32x4*)(p.norm_w + k), w1 = *(const f32x4*)(p.norm_w + k + 4);
                u16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float wj = j < 4 ? w0[j] : w1[j - 4];
                    o[j] = f32_to_bf16(wj * (bf16_to_f32(v[j]) * rstd));
                }
                *(u16x8*)(xs + (size_t)m * xs_stride + k) = o;
            }
        }
        __syncthreads();
    }

    f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    const bf16_t* xl = xs + (size_t)(row_ok ? fr : 0) * xs_stride + g * 16;
    for (int i0 = s0; i0 < s1; i0 += U) {
        if (i0 != s0) issue(i0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = i0 + u;
            if (s < s1) {
                u32x4 x0, x1;
                if (NORM) {
                    const u32x4* px = (const u32x4*)(xl + (size_t)s * 64);
                    x0 = row_ok ? px[0] : (u32x4){0u, 0u, 0u, 0u};
                    x1 = row_ok ? px[1] : (u32x4){0u, 0u, 0u, 0u};
                } else {
                    x0 = qx[u][0]; x1 = qx[u][1];
                }
                u32x4 a0 = qa[u][0], a1 = qa[u][1], b0 = qb[u][0], b1 = qb[u][1];
                if (FP8) {
                    widen_fp8(qa[u][0], a0, a1);
                    if (SWIGLU) widen_fp8(qb[u][0], b0, b1);
                }
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a0), as_bf16x8(x0), acc_a, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a1), as_bf16x8(x1), acc_a, 0, 0, 0);
                if (SWIGLU) {
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b0), as_bf16x8(x0), acc_b, 0, 0, 0);
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b1), as_bf16x8(x1), acc_b, 0, 0, 0);
                }
            }
        }
    }

    // ---- the eight K-slices meet in LDS; wave 0 finishes: lane (m = fr, g) holds outputs n = 4g..4g+3 of row m ----
    *(f32x4*)(red + ((size_t)(wave * 2 + 0) * 64 + lane) * 4) = acc_a;
    if (SWIGLU) *(f32x4*)(red + ((size_t)(wave * 2 + 1) * 64 + lane) * 4) = acc_b;
    __syncthreads();
    if (wave != 0 || !row_ok) return;
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        sa += *(const f32x4*)(red + ((size_t)(w * 2 + 0) * 64 + lane) * 4);
        if (SWIGLU) sb += *(const f32x4*)(red + ((size_t)(w * 2 + 1) * 64 + lane) * 4);
    }
    const int m = fr;
    const int n_out_total = SWIGLU ? p.N / 2 : p.N;
    const int n0 = grp * 16 + g * 4;                  // output column (SwiGLU: group j of 16 outputs = packed rows 32j..32j+31)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= n_out_total) break;
        float t;
        if (SWIGLU) {
            float gt = sa[j], up = sb[j];
            if (FP8) { gt *= p.wscale[grp * 32 + g * 4 + j]; up *= p.wscale[grp * 32 + 16 + g * 4 + j]; }    // the rows' 2^e, once
            t = act_silu(gt) * up;
        } else {
            t = sa[j];
            if (FP8) t *= p.wscale[n];
            if (p.bias) t += p.bias[n];
            t = apply_act(t, p.act);
        }
        if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + n]);
        if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + n] = t;
        else ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(t);
    }
}

// 17..64 activation rows: the same weight stream with MH = 2 or 4 B operands per weight fragment (rows 16h .. 16h+15) - one pass
// over the weights for up to 64 sequences.  Activations come straight from global / L2 (already normalised by the caller: 32+ rows x
// 8 KiB do not fit LDS next to a second workgroup); 4 steps in flight (2 with four operands).  Same k assignment and slice
// order as skinny_kernel.
template <bool SWIGLU, int NW, bool FP8, int MH>
__global__ __launch_bounds__(NW * 64) void skinny_wide_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int K = p.K;
    float* red = (float*)smem;                                             // [NW][2 MH][64][4]

    const int grp = blockIdx.x;
    int row_a, row_b = 0;                                                  // row this lane streams (tile a; SwiGLU: tile b = up)
    if (SWIGLU) { row_a = grp * 32 + fr; row_b = row_a + 16; }
    else { row_a = grp * 16 + fr; row_a = row_a < p.N ? row_a : p.N - 1; }
    const WAddr wad = waddr<SWIGLU>(p, grp, lane);
    const bf16_t* wa = wad.a;
    const bf16_t* wb = wad.b;
    const int sstr = wad.sstr, hoff = wad.hoff;
    const unsigned char* wa8 = p.W8 + (size_t)row_a * p.ldw + g * 16;      // FP8: 16 bytes = the same 16 k of a step
    const unsigned char* wb8 = p.W8 + (size_t)row_b * p.ldw + g * 16;

    constexpr int U = MH == 2 ? 4 : 2;
    const int steps = K >> 6;
    const int per = (steps + NW - 1) / NW;
    const int s0 = wave * per < steps ? wave * per : steps;
    const int s1 = s0 + per < steps ? s0 + per : steps;

    u32x4 qa[U][2], qb[U][2], qx[MH][U][2];
    bool ok[MH];
    const bf16_t* xg[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        ok[h] = fr + 16 * h < p.M;
        xg[h] = p.A + (size_t)(ok[h] ? fr + 16 * h : 0) * p.lda + g * 16;
    }
    auto issue = [&](int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = i0 + u;
            if (s < s1) {
                if (FP8) {
                    qa[u][0] = *(const u32x4*)(wa8 + (size_t)s * 64);
                    if (SWIGLU) qb[u][0] = *(const u32x4*)(wb8 + (size_t)s * 64);
                } else {
                    const bf16_t* pa = wa + (size_t)s * sstr;
                    qa[u][0] = *(const u32x4*)pa; qa[u][1] = *(const u32x4*)(pa + hoff);        // default cache policy (tiled copy: non-temporal measured equal)
                    if (SWIGLU) {
                        const bf16_t* pb = wb + (size_t)s * sstr;
                        qb[u][0] = *(const u32x4*)pb; qb[u][1] = *(const u32x4*)(pb + hoff);
                    }
                }
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    const u32x4* px = (const u32x4*)(xg[h] + (size_t)s * 64);
                    qx[h][u][0] = ok[h] ? px[0] : (u32x4){0u, 0u, 0u, 0u};
                    qx[h][u][1] = ok[h] ? px[1] : (u32x4){0u, 0u, 0u, 0u};
                }
            }
        }
    };
    issue(s0);

    f32x4 acc_a[MH], acc_b[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) { acc_a[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc_b[h] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int i0 = s0; i0 < s1; i0 += U) {
        if (i0 != s0) issue(i0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = i0 + u;
            if (s < s1) {
                u32x4 a0 = qa[u][0], a1 = qa[u][1], b0 = qb[u][0], b1 = qb[u][1];
                if (FP8) {
                    widen_fp8(qa[u][0], a0, a1);
                    if (SWIGLU) widen_fp8(qb[u][0], b0, b1);
                }
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    acc_a[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a0), as_bf16x8(qx[h][u][0]), acc_a[h], 0, 0, 0);
                    acc_a[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a1), as_bf16x8(qx[h][u][1]), acc_a[h], 0, 0, 0);
                    if (SWIGLU) {
                        acc_b[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b0), as_bf16x8(qx[h][u][0]), acc_b[h], 0, 0, 0);
                        acc_b[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b1), as_bf16x8(qx[h][u][1]), acc_b[h], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- the K-slices meet in LDS ([NW][2 MH tiles][64][4]); wave h finishes rows 16h .. 16h+15 ----
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        *(f32x4*)(red + ((size_t)(wave * 2 * MH + h) * 64 + lane) * 4) = acc_a[h];
        if (SWIGLU) *(f32x4*)(red + ((size_t)(wave * 2 * MH + MH + h) * 64 + lane) * 4) = acc_b[h];
    }
    __syncthreads();
    if (wave >= MH) return;
    const int m = fr + 16 * wave;
    if (m >= p.M) return;
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        sa += *(const f32x4*)(red + ((size_t)(w * 2 * MH + wave) * 64 + lane) * 4);
        if (SWIGLU) sb += *(const f32x4*)(red + ((size_t)(w * 2 * MH + MH + wave) * 64 + lane) * 4);
    }
    const int n_out_total = SWIGLU ? p.N / 2 : p.N;
    const int n0 = grp * 16 + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= n_out_total) break;
        float t;
        if (SWIGLU) {
            float gt = sa[j], up = sb[j];
            if (FP8) { gt *= p.wscale[grp * 32 + g * 4 + j]; up *= p.wscale[grp * 32 + 16 + g * 4 + j]; }
            t = act_silu(gt) * up;
        } else {
            t = sa[j];
            if (FP8) t *= p.wscale[n];
            if (p.bias) t += p.bias[n];
            t = apply_act(t, p.act);
        }
        if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + n]);
        if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + n] = t;
        else ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(t);
    }
}

// Persistent form of the fused-norm variant (QKV, gate-up, lm_head of a 2..16-row decode step): one 8-wave workgroup per CU
// stages the normalised activation rows in LDS ONCE and then walks its row groups (blockIdx.x, + gridDim.x, ...), so x is read
// 256 times per launch instead of once per row group (gate-up: 896 groups, lm_head: 2000) and never competes with the weight
// stream for L2.  The weight loads run U steps ahead of the MFMAs ACROSS group boundaries (slot u is refilled right after it is
// consumed, possibly with the next group's first steps), so the reduction / epilogue of a group hides under the loads in flight.
// Same k assignment, same per-wave K-slices and the same summation order over the eight slices as skinny_kernel: results are
// bit-identical to it.  Needs K/64 divisible by 8 waves x U steps (hidden 4096: 64 steps = 8 x 8).
// hipcc waits for ALL loads in flight at the head of every 8-step pass (the refill sits behind a branch, its waitcnt pass does not
// count across the back edge): bursts of 8 steps.  Measured alternative - unconditional refills, `s_waitcnt vmcnt(14)` before
// every slot, seven slots always in flight: gate-up 64.9 us instead of 58.8 at 16 rows.  The synchronised bursts of the eight
// waves (whole 8 KiB rows requested together) serve the DRAM better than eight desynchronised streams; kept as is.
template <bool SWIGLU, bool FP8>
__global__ __launch_bounds__(512) void skinny_persist_kernel(SkinnyParams p, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8;
    constexpr int U = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int K = p.K;
    const int xs_stride = K + 8;
    bf16_t* xs = (bf16_t*)smem;                                           // [M][K + 8]
    float* red = (float*)(smem + (size_t)p.M * xs_stride * 2);            // [NW][2][64][4]
    const int per = (K >> 6) / NW;                                        // steps per wave and group (host: exact, multiple of U)
    const int s0 = wave * per;
    const int G = (n_groups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // my groups (grid <= n_groups)
    const bool row_ok = fr < p.M;

    u32x4 qa[U][2], qb[U][2];
    int pj = 0, ps = 0;                                                   // prefetch cursor: my group ordinal, step within the slice
    auto refill = [&](int u) {
        if (pj < G) {
            const int grp = blockIdx.x + pj * gridDim.x;
            int row_a, row_b = 0;
            if (SWIGLU) { row_a = grp * 32 + fr; row_b = row_a + 16; }
            else { row_a = grp * 16 + fr; row_a = row_a < p.N ? row_a : p.N - 1; }
            const size_t koff = (size_t)(s0 + ps) * 64 + g * 16;
            if (FP8) {
                qa[u][0] = *(const u32x4*)(p.W8 + (size_t)row_a * p.ldw + koff);
                if (SWIGLU) qb[u][0] = *(const u32x4*)(p.W8 + (size_t)row_b * p.ldw + koff);
            } else {
                const WAddr wad = waddr<SWIGLU>(p, grp, lane);
                const size_t so = (size_t)(s0 + ps) * wad.sstr;
                qa[u][0] = *(const u32x4*)(wad.a + so); qa[u][1] = *(const u32x4*)(wad.a + so + wad.hoff);        // default cache policy (non-temporal: 62.5 vs 57.8 us on gate-up at 16 rows)
                if (SWIGLU) { qb[u][0] = *(const u32x4*)(wad.b + so); qb[u][1] = *(const u32x4*)(wad.b + so + wad.hoff); }
            }
            if (++ps == per) { ps = 0; ++pj; }
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) refill(u);          // the first weights are in flight before the norm prologue

    for (int m = wave; m < p.M; m += NW) {          // one wave per activation row: bf16(norm_w * x * rstd) -> LDS
        const bf16_t* x = p.A + (size_t)m * p.lda;
        float ss = 0.f;
        for (int k = lane * 8; k < K; k += 64 * 8) {
            const u16x8 v = *(const u16x8*)(x + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(v[j]); ss += f * f; }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)K + p.norm_eps);
        for (int k = lane * 8; k < K; k += 64 * 8) {
            const u16x8 v = *(const u16x8*)(x + k);
            const f32x4 w0 = *(const f32x4*)(p.norm_w + k), w1 = *(const f32x4*)(p.norm_w + k + 4);
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wj = j < 4 ? w0[j] : w1[j - 4];
                o[j] = f32_to_bf16(wj * (bf16_to_f32(v[j]) * rstd));
            }
            *(u16x8*)(xs + (size_t)m * xs_stride + k) = o;
        }
    }
    __syncthreads();

    const bf16_t* xl = xs + (size_t)(row_ok ? fr : 0) * xs_stride + g * 16;
    const int n_out_total = SWIGLU ? p.N / 2 : p.N;
    for (int j = 0; j < G; ++j) {
        f32x4 acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = 0; i0 < per; i0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32x4* px = (const u32x4*)(xl + (size_t)(s0 + i0 + u) * 64);
                const u32x4 x0 = row_ok ? px[0] : (u32x4){0u, 0u, 0u, 0u};
                const u32x4 x1 = row_ok ? px[1] : (u32x4){0u, 0u, 0u, 0u};
                u32x4 a0 = qa[u][0], a1 = qa[u][1], b0 = qb[u][0], b1 = qb[u][1];
                if (FP8) {
                    widen_fp8(qa[u][0], a0, a1);
                    if (SWIGLU) widen_fp8(qb[u][0], b0, b1);
                }
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a0), as_bf16x8(x0), acc_a, 0, 0, 0);
                acc_a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a1), as_bf16x8(x1), acc_a, 0, 0, 0);
                if (SWIGLU) {
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b0), as_bf16x8(x0), acc_b, 0, 0, 0);
                    acc_b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(b1), as_bf16x8(x1), acc_b, 0, 0, 0);
                }
                refill(u);                          // this slot's next step, U ahead (the next group's once this one is issued out)
            }
        }
        // ---- the eight K-slices meet in LDS; wave 0 sums them (slice order 0..7) and finishes the group ----
        *(f32x4*)(red + ((size_t)(wave * 2 + 0) * 64 + lane) * 4) = acc_a;
        if (SWIGLU) *(f32x4*)(red + ((size_t)(wave * 2 + 1) * 64 + lane) * 4) = acc_b;
        __syncthreads();
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                sa += *(const f32x4*)(red + ((size_t)(w * 2 + 0) * 64 + lane) * 4);
                if (SWIGLU) sb += *(const f32x4*)(red + ((size_t)(w * 2 + 1) * 64 + lane) * 4);
            }
        }
        __syncthreads();                            // red is free for the next group; wave 0 finishes from registers
        if (wave == 0 && row_ok) {
            const int grp = blockIdx.x + j * gridDim.x;
            const int m = fr;
            const int n0 = grp * 16 + g * 4;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = n0 + jj;
                if (n >= n_out_total) break;
                float t;
                if (SWIGLU) {
                    float gt = sa[jj], up = sb[jj];
                    if (FP8) { gt *= p.wscale[grp * 32 + g * 4 + jj]; up *= p.wscale[grp * 32 + 16 + g * 4 + jj]; }
                    t = act_silu(gt) * up;
                } else {
                    t = sa[jj];
                    if (FP8) t *= p.wscale[n];
                    if (p.bias) t += p.bias[n];
                    t = apply_act(t, p.act);
                }
                if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + n]);
                if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + n] = t;
                else ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(t);
            }
        }
    }
}

size_t skinny_persist_lds(const LinearArgs& a) { return (size_t)a.M * (a.K + 8) * 2 + (size_t)8 * 2 * 64 * 4 * sizeof(float); }

bool skinny_persist_ok(const LinearArgs& a) {
    if (!a.norm_w) return false;
    const int steps = a.K >> 6, U = 8;
    return steps % 8 == 0 && (steps / 8) % U == 0 && skinny_persist_lds(a) <= 160 * 1024;
}

int g_num_cu_skinny = 0;

template <bool SWIGLU, bool FP8>
int launch_persist(const SkinnyParams& p, int n_groups, size_t lds, hipStream_t s) {
    // ONE workgroup per CU even where LDS would hold two (5..7 rows): measured 3.85 / 3.88 / 3.94 ms per 5 / 6 / 7-row step
    // against 3.98 / 4.03 / 4.08 with two (knob 9 = 4) - the eight waves' synchronised bursts again
    const int per_cu = g_skinny_mode == 4 ? ((160 * 1024) / (int)lds >= 2 ? 2 : 1) : 1;
    const int cap = g_num_cu_skinny * per_cu;
    // the launch lasts as long as its busiest workgroup: ceil(n_groups / cap) groups.  With that many groups on EVERY workgroup fewer of them
    // are needed (QKV: 384 groups = 2 per workgroup on 192 CUs instead of 2 on 128 + 1 on 128; the waves' bursts then share HBM with fewer
    // others): same summation per group, bit-identical
    const int rounds = (n_groups + cap - 1) / cap;
    const int grid = g_skinny_even ? (n_groups + rounds - 1) / rounds : (n_groups < cap ? n_groups : cap);
    vz_launch_timed(skinny_persist_kernel<SWIGLU, FP8>, dim3(grid), dim3(512), lds, s, p, n_groups);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

size_t skinny_lds(const LinearArgs& a) {
    return (a.norm_w ? (size_t)a.M * (a.K + 8) * 2 : 0) + (size_t)8 * 2 * 64 * 4 * sizeof(float);
}

template <bool SWIGLU, bool NORM, int NW, bool FP8>
int launch_nw(const SkinnyParams& p, int blocks, size_t lds, hipStream_t s) {
    static VzDeviceOnce attr;
    if (vz_device_first(attr)) {
        VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_kernel<SWIGLU, NORM, NW, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    vz_launch_timed(skinny_kernel<SWIGLU, NORM, NW, FP8>, dim3(blocks), dim3(NW * 64), lds, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

template <bool SWIGLU, bool NORM>
int launch(const SkinnyParams& p, int blocks, size_t lds, hipStream_t s) {
    const bool four = blocks >= 512 && (p.K >> 6) >= 16;
    if (p.W8) return four ? launch_nw<SWIGLU, NORM, 4, true>(p, blocks, lds, s) : launch_nw<SWIGLU, NORM, 8, true>(p, blocks, lds, s);
    return four ? launch_nw<SWIGLU, NORM, 4, false>(p, blocks, lds, s) : launch_nw<SWIGLU, NORM, 8, false>(p, blocks, lds, s);
}

}  // namespace

int g_skinny_mode = 1;   // vz_tune_set(9, v): 1 = 2..16-row linears use the MFMA weight stream (default), 0 = GEMV / tile GEMM as before

// 5..16 rows with the RMSNorm fused: only the persistent form holds the normalised rows in LDS at one workgroup per CU
bool vz_skinny_fused_norm_ok(const LinearArgs& a) { return g_skinny_mode != 2 && a.M >= 2 && a.M <= 16 && skinny_persist_ok(a); }

bool vz_skinny_ok(const LinearArgs& a) {
    if (a.M < 2 || a.M > 64 || (a.K & 63) != 0 || a.K < 512) return false;
    if (a.M > 16 && (a.norm_w || !a.wide_ok)) return false;                                       // 17..64 rows: the caller normalises (no LDS staging)
    if (a.W8 && (!a.wscale || (a.ldw & 15) != 0 || ((uintptr_t)a.W8 & 15) != 0)) return false;
    if (a.act == VZ_ACT_SWIGLU && (a.N % 32) != 0) return false;
    if ((a.lda & 7) != 0 || (a.ldw & 7) != 0) return false;                       // 16-byte fragment loads
    return skinny_persist_ok(a) || skinny_lds(a) <= 160 * 1024;
}

int vz_init_skinny_kernels() {
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    // every variant a captured decode step can reach gets its dynamic-LDS limit now (never inside a stream capture)
#define VZ_SK_ATTR(SW, NM, W)                                                                                                                   \
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_kernel<SW, NM, W, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_kernel<SW, NM, W, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_SK_ATTR(false, false, 4) VZ_SK_ATTR(false, false, 8) VZ_SK_ATTR(false, true, 4) VZ_SK_ATTR(false, true, 8)
    VZ_SK_ATTR(true, false, 4) VZ_SK_ATTR(true, false, 8) VZ_SK_ATTR(true, true, 4) VZ_SK_ATTR(true, true, 8)
#undef VZ_SK_ATTR
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_persist_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_persist_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_persist_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_persist_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    {
        int dev = 0;
        hipDeviceProp_t prop;
        VZ_CHECK_HIP(hipGetDevice(&dev));
        VZ_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        g_num_cu_skinny = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return VZ_OK;
}

int vz_launch_tile_weights(const bf16_t* W, int N, int K, int ldw, bf16_t* Wt, hipStream_t s) {
    VZ_CHECK_ARG(W && Wt && N > 0 && (N & 15) == 0 && K >= 64 && (K & 63) == 0 && ldw >= K && (ldw & 7) == 0 &&
                 ((uintptr_t)W & 15) == 0 && ((uintptr_t)Wt & 15) == 0, "tile_weights: needs N %% 16 == 0, K %% 64 == 0, 16-byte-aligned rows (N=%d K=%d)", N, K);
    const long chunks = (long)(N >> 4) * (K >> 6) * 128;
    const int blocks = (int)std::min<long>((chunks + 255) / 256, 8192);
    hipLaunchKernelGGL(tile_weights_kernel, dim3(blocks), dim3(256), 0, s, W, ldw, Wt, N, K);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_skinny(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(vz_skinny_ok(a), "skinny gemm: needs 2 <= M <= 64 (fused norm: <= 16), bf16 weights, K %% 64 == 0 and >= 512, 16-byte-aligned rows (M=%d K=%d)", a.M, a.K);
    { int r = vz_init_skinny_kernels(); if (r) return r; }
    SkinnyParams p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual; p.norm_w = a.norm_w;
    p.W8 = a.W8; p.wscale = a.wscale;
    p.Wt = (!a.W8 && (a.N & 15) == 0 && g_skinny_mode != 5) ? a.Wt : nullptr;        // (knob 9 = 5: ignore the tiled copies, A/B)
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32; p.norm_eps = a.norm_eps;
    const bool sw = a.act == VZ_ACT_SWIGLU;
    const int blocks = sw ? a.N / 32 : (a.N + 15) / 16;
    if (a.M > 16) {          // 17..64 rows: two / four B operands per weight fragment
        const bool four = blocks >= 512 && (p.K >> 6) >= 16;
#define VZ_WIDE(SW, NWV, F8, MHV) do { \
            static VzDeviceOnce attr; \
            if (vz_device_first(attr)) { VZ_CHECK_HIP(hipFuncSetAttribute((const void*)skinny_wide_kernel<SW, NWV, F8, MHV>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); } \
            vz_launch_timed(skinny_wide_kernel<SW, NWV, F8, MHV>, dim3(blocks), dim3(NWV * 64), (size_t)NWV * 2 * MHV * 64 * 4 * sizeof(float), s, p); \
            VZ_LAUNCH_CHECK(); return VZ_OK; } while (0)
#define VZ_WIDE_MH(SW, NWV, F8) do { if (a.M > 32) VZ_WIDE(SW, NWV, F8, 4); VZ_WIDE(SW, NWV, F8, 2); } while (0)
        if (p.W8) { if (sw) { if (four) VZ_WIDE_MH(true, 4, true); VZ_WIDE_MH(true, 8, true); } if (four) VZ_WIDE_MH(false, 4, true); VZ_WIDE_MH(false, 8, true); }
        if (sw) { if (four) VZ_WIDE_MH(true, 4, false); VZ_WIDE_MH(true, 8, false); }
        if (four) VZ_WIDE_MH(false, 4, false);
        VZ_WIDE_MH(false, 8, false);
#undef VZ_WIDE_MH
#undef VZ_WIDE
    }
    if (g_skinny_mode != 2 && skinny_persist_ok(a) && (a.M >= 3 || g_skinny_mode == 3)) {      // knob 9: 2 = never (3..4 rows joined in round 2: 3.59 / 3.61 vs 3.69 / 3.68 ms per step on the tiled weights)
        const size_t pl = skinny_persist_lds(a);
        if (p.W8) return sw ? launch_persist<true, true>(p, blocks, pl, s) : launch_persist<false, true>(p, blocks, pl, s);
        return sw ? launch_persist<true, false>(p, blocks, pl, s) : launch_persist<false, false>(p, blocks, pl, s);
    }
    const size_t lds = skinny_lds(a);
    VZ_CHECK_ARG(lds <= 160 * 1024, "skinny gemm: %d activation rows of K=%d do not fit the LDS staging", a.M, a.K);
    if (sw) return a.norm_w ? launch<true, true>(p, blocks, lds, s) : launch<true, false>(p, blocks, lds, s);
    return a.norm_w ? launch<false, true>(p, blocks, lds, s) : launch<false, false>(p, blocks, lds, s);
}
