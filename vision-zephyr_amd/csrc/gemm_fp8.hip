// FP8 MFMA prefill for the W8A16 engine (SURVEY config 5 "fp8 MFMA weights"; gfx950): C[M,N'] = epi(x8[M,K] . W8[N,K]^T * sx[m] * sw[n]).
//
//   * weights: the engine's e4m3 copies with ONE power-of-two scale per output row (vz_hip/quant.py) - unchanged;
//   * activations: quant_rows_fp8_kernel gives every row of the bf16 input ONE power-of-two scale 2^e, e = the smallest integer with
//     max|x_row| <= 448 * 2^e, and rounds x * 2^-e to OCP e4m3 (round-to-nearest-even) - the same quantiser as the weights'
//     (vz_hip/quant.py::quantize_rows, which the oracle restates), so parity is against a reference that quantises identically;
//   * the product runs on v_mfma_scale_f32_16x16x128_f8f6f4 with both block scales = 2^0 (an e4m3 x e4m3 product at twice the bf16
//     rate per clock); the two row scales multiply the fp32 sum once, in the epilogue, like the W8A16 GEMV does.
// Kernel = gemm.hip's 128 x 128 tile (4 waves x 64 x 64, 16-byte LDS-DMA into XOR-swizzled double buffers, 2 workgroups per CU) with a
// K-tile of 128 BYTES = 128 k: the LDS image, the staging and the swizzle are byte-for-byte those of the bf16 kernel's 64-k tile; a
// 16x16x128 fragment is 32 contiguous bytes of a row per lane (two ds_read_b128), 16 MFMAs per K-tile and wave instead of 64.
// Both MFMA operands use the same lane -> k mapping (row / column = lane & 15, k bytes 32 (lane >> 4) .. + 31), which is all the
// contraction needs.
#include "vz_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int BM = 128, BN = 128, BKB = 128;          // K-tile in bytes (= k)
constexpr int TILE_BYTES = BM * BKB;                  // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int FP8_LDS = 2 * BUF_BYTES;                // 64 KiB: 2 workgroups per CU

struct Fp8Params {
    const unsigned char* A8; const float* ascale; const unsigned char* W8; const float* wscale; void* C;
    const float* bias; const bf16_t* residual;
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32, tiles_m, tiles_n;
};

__device__ __forceinline__ void glds16(const unsigned char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// ---- per-row e4m3 quantisation of bf16 activations: one wave per row, two passes over the (L2-resident) row ----
// e = x - 8 + (m > 1.75) for max|x| = m * 2^x, m in [1, 2): the smallest e with max|x| <= 448 * 2^e (448 = 1.75 * 2^8); no logarithm.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ q, int ldq,
                                                             float* __restrict__ scale, int rows, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    float amax = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const u16x8 v = *(const u16x8*)(xr + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf16_to_f32(v[j])));
    }
    amax = wave_max(amax);
    int e = 0;
    if (amax > 0.f) {
        const unsigned bits = __float_as_uint(amax);
        const int ex = (int)((bits >> 23) & 0xff) - 127;
        const unsigned man = bits & 0x7fffffu;
        e = ex - 8 + (man > 0x600000u ? 1 : 0);            // mantissa of 1.75 = 0x600000
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);     // 2^-e
    if (lane == 0) scale[row] = __uint_as_float((unsigned)(127 + e) << 23);
    unsigned char* qr = q + (size_t)row * ldq;
    for (int k = lane * 8; k < K; k += 512) {
        const u16x8 v = *(const u16x8*)(xr + k);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(v[0]) * inv, bf16_to_f32(v[1]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(v[2]) * inv, bf16_to_f32(v[3]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(v[4]) * inv, bf16_to_f32(v[5]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(v[6]) * inv, bf16_to_f32(v[7]) * inv, hi, true);
        *(uint2*)(qr + k) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// Whole-chunk widths (K = NCH * 512: Zephyr's 4096 and 14336): the row is requested in one go - NCH loads in flight per lane, where the
// loop above waits for each 16-byte piece before it asks for the next - and quantised from registers (one pass over the row, not two).
// Same arithmetic: the same bytes and scales.
template <int NCH>
__global__ __launch_bounds__(256) void quant_rows_fp8_rows_kernel(const bf16_t* __restrict__ x, int ldx, unsigned char* __restrict__ q, int ldq,
                                                                  float* __restrict__ scale, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx + lane * 8;
    u16x8 t[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) t[c] = *(const u16x8*)(xr + c * 512);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf16_to_f32(t[c][j])));
    amax = wave_max(amax);
    int e = 0;
    if (amax > 0.f) {
        const unsigned bits = __float_as_uint(amax);
        e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);     // 2^-e
    if (lane == 0) scale[row] = __uint_as_float((unsigned)(127 + e) << 23);
    unsigned char* qr = q + (size_t)row * ldq + lane * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(t[c][0]) * inv, bf16_to_f32(t[c][1]) * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(t[c][2]) * inv, bf16_to_f32(t[c][3]) * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(t[c][4]) * inv, bf16_to_f32(t[c][5]) * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16_to_f32(t[c][6]) * inv, bf16_to_f32(t[c][7]) * inv, hi, true);
        *(uint2*)(qr + c * 512) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_quant_fp8_rows_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w, float eps,
                                                                     unsigned char* __restrict__ q, int ldq, float* __restrict__ scale, int rows) {
    constexpr int cols = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx + lane * 8;
    u16x8 t[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) t[c] = *(const u16x8*)(xr + c * 512);
    const float* wr = w + lane * 8;
    f32x4 w0[NCH], w1[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { w0[c] = *(const f32x4*)(wr + c * 512); w1[c] = *(const f32x4*)(wr + c * 512 + 4); }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[c][j] = bf16_to_f32(t[c][j]); s += v[c][j] * v[c][j]; }
    s = wave_sum(s);
    const float rstd = rsqrtf(s / (float)cols + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wj = j < 4 ? w0[c][j] : w1[c][j - 4];
            v[c][j] = bf16_to_f32(f32_to_bf16(wj * (v[c][j] * rstd)));
            amax = fmaxf(amax, fabsf(v[c][j]));
        }
    amax = wave_max(amax);
    int e = 0;
    if (amax > 0.f) {
        const unsigned bits = __float_as_uint(amax);
        e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);
    if (lane == 0) scale[row] = __uint_as_float((unsigned)(127 + e) << 23);
    unsigned char* qr = q + (size_t)row * ldq + lane * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
        *(uint2*)(qr + c * 512) = make_uint2((unsigned)lo, (unsigned)hi);
    }
}

// RMSNorm (hf:models/mistral/modeling_mistral.py:182-199) + the quantiser above in one pass: the normalised row is rounded to bf16 exactly
// as norm_kernel stores it (w * (x * rstd)), then to e4m3 - the same bytes and scale as the two launches (tested), without the bf16 round trip
// through HBM.  One wave per row, the row in registers (cols <= 5120).
__global__ __launch_bounds__(256) void rmsnorm_quant_fp8_kernel(const bf16_t* __restrict__ x, int ldx, const float* __restrict__ w, float eps,
                                                                unsigned char* __restrict__ q, int ldq, float* __restrict__ scale, int rows, int cols) {
    constexpr int MAXC = 10;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    const int nch = (cols + 511) >> 9;
    float v[MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 512 + lane * 8;
        if (c < nch && k < cols) {
            const u16x8 t = *(const u16x8*)(xr + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = bf16_to_f32(t[j]); s += v[c][j] * v[c][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    s = wave_sum(s);
    const float rstd = rsqrtf(s / (float)cols + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 512 + lane * 8;
        if (c < nch && k < cols) {
            const f32x4 w0 = *(const f32x4*)(w + k), w1 = *(const f32x4*)(w + k + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wj = j < 4 ? w0[j] : w1[j - 4];
                v[c][j] = bf16_to_f32(f32_to_bf16(wj * (v[c][j] * rstd)));
                amax = fmaxf(amax, fabsf(v[c][j]));
            }
        }
    }
    amax = wave_max(amax);
    int e = 0;
    if (amax > 0.f) {
        const unsigned bits = __float_as_uint(amax);
        e = (int)((bits >> 23) & 0xff) - 127 - 8 + ((bits & 0x7fffffu) > 0x600000u ? 1 : 0);
        e = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
    const float inv = __uint_as_float((unsigned)(127 - e) << 23);
    if (lane == 0) scale[row] = __uint_as_float((unsigned)(127 + e) << 23);
    unsigned char* qr = q + (size_t)row * ldq;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int k = c * 512 + lane * 8;
        if (c < nch && k < cols) {
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, hi, true);
            *(uint2*)(qr + k) = make_uint2((unsigned)lo, (unsigned)hi);
        }
    }
}

// Fast epilogue (all 128 columns of the tile inside N, vectorisable rows): activation, bias / residual presence and output type are
// decided once per tile, the weight scales and bias of the lane's 16 columns are loaded once; the generic path below tests every
// element (gemm.hip measured that form at 29 us against 5 us per tile).  Same arithmetic per element: bit-identical results.
__device__ __forceinline__ void fp8_put4(const Fp8Params& p, bool has_res, bool f32, int m, int n0, f32x4 v) {
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

template <int ACT>
__device__ __forceinline__ void fp8_epilogue_fast(const Fp8Params& p, f32x4 (&acc)[4][4], int m_base, int n_base, int n_half) {
    const bool has_res = p.residual != nullptr, f32 = p.out_fp32 != 0;
    f32x4 ws[4], b4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        ws[nt] = *(const f32x4*)(p.wscale + n_base + nt * 16);
        b4[nt] = (ACT != VZ_ACT_SWIGLU && p.bias) ? *(const f32x4*)(p.bias + n_base + nt * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const bool has_bias = ACT != VZ_ACT_SWIGLU && p.bias != nullptr;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m_base + mt * 16;
        if (m >= p.M) continue;
        const float sx = p.ascale[m];
        if constexpr (ACT == VZ_ACT_SWIGLU) {
#pragma unroll
            for (int nt = 0; nt < 4; nt += 2) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gt = acc[nt][mt][j] * (sx * ws[nt][j]);
                    const float up = acc[nt + 1][mt][j] * (sx * ws[nt + 1][j]);
                    v[j] = act_silu(gt) * up;
                }
                fp8_put4(p, has_res, f32, m, n_half + (nt >> 1) * 16, v);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[nt][mt][j] * (sx * ws[nt][j]);
                    if (has_bias) t += b4[nt][j];
                    v[j] = ACT == VZ_ACT_QUICK_GELU ? act_quick_gelu(t) : (ACT == VZ_ACT_GELU_ERF ? act_gelu_erf(t) : t);
                }
                fp8_put4(p, has_res, f32, m, n_base + nt * 16, v);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(Fp8Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, 64(m) x 64(n) each

    // XCD-aware, bijective tile order (gemm.hip): blocks b and b+8 share an XCD; inside a run tiles walk M first
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bn = tile / p.tiles_m, bm = tile - bn * p.tiles_m;

    // ---- staging addresses: 1024 16-byte chunks per operand tile, 4 per thread ----
    const unsigned char* ga[4];
    const unsigned char* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = i * 256 + tid;
        const int row = ch >> 3, c = ch & 7;
        const int gc = c ^ (row & 7);  // swizzle on the source; the LDS image stays lane-linear
        int arow = bm * BM + row; arow = arow < p.M ? arow : p.M - 1;
        int wrow = bn * BN + row; wrow = wrow < p.N ? wrow : p.N - 1;
        ga[i] = p.A8 + (size_t)arow * p.lda + gc * 16;
        gw[i] = p.W8 + (size_t)wrow * p.ldw + gc * 16;
    }
    const int wave_chunk = wave * 64 * 16;

    // ---- fragment read addresses (16x16x128: lane holds row lane&15, k bytes 32 g .. 32 g + 31 = chunks 2g, 2g+1) ----
    const int frow = lane & 15, g = lane >> 4;
    const int c0 = ((2 * g) ^ (frow & 7)) << 4, c1 = ((2 * g + 1) ^ (frow & 7)) << 4;
    const int a_off = (wm * 64 + frow) * 128;
    const int w_off = TILE_BYTES + (wn * 64 + frow) * 128;

    f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BKB;
    const int a_slices = min(4, (p.M - bm * BM + 31) >> 5);       // activation rows past M are never stored: their slices are not loaded
    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * BUF_BYTES + wave_chunk;
        char* lw = la + TILE_BYTES;
        const int kb = kt * BKB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < a_slices) glds16(ga[i] + kb, la + i * 4096);
            glds16(gw[i] + kb, lw + i * 4096);
        }
    };

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    const int one = 0x7f7f7f7f;         // e8m0 127 = 2^0 in every byte: both block scales are 1
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* base = smem + cur * BUF_BYTES;
        i32x8 af[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32x4 a0 = *(const u32x4*)(base + a_off + t * 2048 + c0), a1 = *(const u32x4*)(base + a_off + t * 2048 + c1);
            const u32x4 w0 = *(const u32x4*)(base + w_off + t * 2048 + c0), w1 = *(const u32x4*)(base + w_off + t * 2048 + c1);
            af[t] = (i32x8){(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
            wf[t] = (i32x8){(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], acc[nt][mt], 0, 0, 0, one, 0, one);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: acc[nt][mt][j] = sum for C[m = .. + mt*16 + (lane&15)][n = .. + nt*16 + 4*(lane>>4) + j] ----
    const int m_base = bm * BM + wm * 64 + frow;
    const int n_base = bn * BN + wn * 64 + g * 4;
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int n_out_total = swiglu ? p.N / 2 : p.N;
    const bool vec_ok = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0);
    if (bn * BN + BN <= p.N && vec_ok && (((uintptr_t)p.bias | (uintptr_t)p.wscale) & 15) == 0) {
        const int n_half = ((bn * BN + wn * 64) >> 1) + g * 4;
        switch (p.act) {
            case VZ_ACT_QUICK_GELU: fp8_epilogue_fast<VZ_ACT_QUICK_GELU>(p, acc, m_base, n_base, n_half); break;
            case VZ_ACT_GELU_ERF: fp8_epilogue_fast<VZ_ACT_GELU_ERF>(p, acc, m_base, n_base, n_half); break;
            case VZ_ACT_SWIGLU: fp8_epilogue_fast<VZ_ACT_SWIGLU>(p, acc, m_base, n_base, n_half); break;
            default: fp8_epilogue_fast<VZ_ACT_NONE>(p, acc, m_base, n_base, n_half); break;
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m_base + mt * 16;
        if (m >= p.M) continue;
        const float sx = p.ascale[m];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float v[4];
            int n0;
            if (swiglu) {
                if (nt & 1) continue;
                // weight rows are interleaved [16 gate | 16 up]: tile nt = gate, nt+1 = up, same lane slots
                const int ng = n_base + nt * 16;                         // gate rows ng .. ng+3, up rows ng+16 ..
                n0 = ((bn * BN + wn * 64) >> 1) + (nt >> 1) * 16 + g * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float gt = acc[nt][mt][j] * (sx * p.wscale[ng + j]);
                    const float up = acc[nt + 1][mt][j] * (sx * p.wscale[ng + 16 + j]);
                    v[j] = act_silu(gt) * up;
                }
            } else {
                n0 = n_base + nt * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = n0 + j < p.N ? acc[nt][mt][j] * (sx * p.wscale[n0 + j]) : 0.f;
                    if (p.bias && n0 + j < p.N) t += p.bias[n0 + j];
                    v[j] = apply_act(t, p.act);
                }
            }
            if (n0 >= n_out_total) continue;
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
    }
}

}  // namespace

int g_fp8_gemm_choice = 0;      // vz_tune_set(21, v): 0 = by grid size, 1 = always the 128^2 kernel, 2 = always the 256^2 pipeline

int vz_launch_quant_rows_fp8(const bf16_t* x, int ldx, unsigned char* q, int ldq, float* scale, int rows, int K, hipStream_t s) {
    VZ_CHECK_ARG(x && q && scale && rows > 0 && K > 0 && (K & 7) == 0 && (ldx & 7) == 0 && (ldq & 7) == 0 && ldx >= K && ldq >= K,
                 "quant_rows_fp8: K, ldx, ldq must be multiples of 8 (K=%d)", K);
    if (K == 4096) hipLaunchKernelGGL((quant_rows_fp8_rows_kernel<8>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, q, ldq, scale, rows);
    else if (K == 14336) hipLaunchKernelGGL((quant_rows_fp8_rows_kernel<28>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, q, ldq, scale, rows);
    else hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, q, ldq, scale, rows, K);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_rmsnorm_quant_fp8(const bf16_t* x, int ldx, const float* w, float eps, unsigned char* q, int ldq, float* scale, int rows, int cols,
                                hipStream_t s) {
    VZ_CHECK_ARG(x && w && q && scale && rows > 0 && (cols & 7) == 0 && cols <= 5120 && (ldx & 7) == 0 && (ldq & 7) == 0,
                 "rmsnorm_quant_fp8: cols=%d must be a multiple of 8 and <= 5120", cols);
    if (cols == 4096) hipLaunchKernelGGL((rmsnorm_quant_fp8_rows_kernel<8>), dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, eps, q, ldq, scale, rows);
    else hipLaunchKernelGGL(rmsnorm_quant_fp8_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, ldx, w, eps, q, ldq, scale, rows, cols);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

bool vz_gemm_fp8_ok(int M, int N, int K, int lda, int ldw) {
    return M > 0 && N > 0 && K >= 128 && (K & 127) == 0 && (lda & 15) == 0 && (ldw & 15) == 0 && lda >= K && ldw >= K;
}

int vz_launch_gemm_fp8(const Fp8LinearArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(a.A8 && a.ascale && a.W8 && a.wscale && a.C, "gemm_fp8: null pointer");
    VZ_CHECK_ARG(vz_gemm_fp8_ok(a.M, a.N, a.K, a.lda, a.ldw), "gemm_fp8: needs K %% 128 == 0 and 16-byte-aligned rows (M=%d N=%d K=%d)", a.M, a.N, a.K);
    VZ_CHECK_ARG(((uintptr_t)a.A8 & 15) == 0 && ((uintptr_t)a.W8 & 15) == 0 && ((uintptr_t)a.C & 15) == 0, "gemm_fp8: pointers must be 16-byte aligned");
    VZ_CHECK_ARG(a.act >= 0 && a.act <= 3 && (a.act != VZ_ACT_SWIGLU || ((a.N & 31) == 0 && !a.bias)), "gemm_fp8: bad activation / SwiGLU shape");
    // grids that fill the chip with 256^2 tiles run on gemm256.hip's deep pipeline (same bytes per K-tile as its bf16 form, twice the FLOPs);
    // knob 21 = 1 keeps everything on the two-stage 128^2 kernel below (A/B)
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    // measured at M = 2048 (tools/bench_fp8.py): QKV (192 tiles) 70.5 vs 99.9 us, gate|up (896) 291 vs 314; O (128 tiles) 65.4 vs 58.5, down (128) 130 vs 128
    if (g_fp8_gemm_choice != 1 && (g_fp8_gemm_choice == 2 || t256 >= 160)) return vz_launch_gemm256_fp8(a, s);
    static VzDeviceOnce attr;
    if (vz_device_first(attr)) {
        VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_fp8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FP8_LDS));
    }
    Fp8Params p;
    p.A8 = a.A8; p.ascale = a.ascale; p.W8 = a.W8; p.wscale = a.wscale; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr; p.act = a.act; p.out_fp32 = a.out_fp32;
    p.tiles_m = (a.M + BM - 1) / BM; p.tiles_n = (a.N + BN - 1) / BN;
    vz_launch_timed(gemm_fp8_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), FP8_LDS, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
