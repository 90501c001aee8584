// Weight-streaming GEMV for the decode step (gfx950): C[M<=8, N'] = epi(x[M,K] . W[N,K]^T).
//
// Batch-1 decode reads every one of the 7.2 B bf16 weights once per token (14.2 GB): the kernel is a
// pure HBM stream.  Each wave owns R weight rows at a time and pulls them with 16-byte loads
// straight into VGPRs (64 lanes x 16 B = 1 KiB contiguous per instruction, U x R instructions in
// flight per wave) - no LDS round trip for data that is read exactly once (cdna guide section 5,
// "GEMV / M <= 16" row); optional non-temporal policy so the stream does not evict what IS reused.
// The activation vector(s) are staged once per workgroup in LDS as bf16; with `norm_w` the
// RMSNorm of the residual stream (hf:models/mistral/modeling_mistral.py:182-199) is fused into
// that staging, and the first weight loads of the workgroup are issued BEFORE the norm so the
// HBM latency of the first tile hides under it.  Accumulation: v_dot2c_f32_bf16 into fp32, wave
// reduction, then the same epilogues as the tile GEMM (bias / act / SwiGLU pair / residual /
// bf16|fp32 out).
//
// FP8 weights (W8A16, SURVEY config 5 "fp8 weights"): the same stream at 1 byte per weight.  Rows are OCP e4m3 with one
// power-of-two scale per output row (vz_hip/quant.py), so the dequantised weight 2^e * fp8 is EXACTLY a bf16 number: a
// lane's 16 weights per load are widened to packed bf16 pairs (v_cvt_scalef32_pk_bf16_fp8, exact) and go
// through the same v_dot2c_f32_bf16 as the bf16 stream; the row's 2^e multiplies the fp32 sum once.  The prefill GEMMs
// run on the bf16 copy of the same dequantised weights, so prefill and decode see one and the same model.
#include "vz_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct GemvParams {
    const bf16_t* A; const bf16_t* W; void* C;
    const unsigned char* W8; const float* wscale;     // FP8 instantiation: e4m3 rows [N][ldw] + fp32 2^e per row
    const float* bias; const bf16_t* residual; const float* norm_w;
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32, units;
    float norm_eps;
};

// NOTE (hipcc / ROCm 7.2): __builtin_bit_cast applied directly to a vector ELEMENT expression (w[i]) folds every
// use to element 0; the elements are copied to scalars first.
__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w0), __builtin_bit_cast(bf16x2, x0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w1), __builtin_bit_cast(bf16x2, x1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w2), __builtin_bit_cast(bf16x2, x2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w3), __builtin_bit_cast(bf16x2, x3), acc, false);
    return acc;
}

// 16 e4m3 weights of one lane -> 16 bf16 (two u32x4 of packed pairs, k order preserved); exact.
// v_cvt_scalef32_pk_bf16_fp8 widens two fp8 of a dword half to a packed bf16 pair in one instruction (scale 1.0).
__device__ __forceinline__ unsigned fp8x2_to_bf16x2(unsigned w, bool hi_half) {
    return hi_half ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                   : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
}
__device__ __forceinline__ void fp8x16_to_bf16(const u32x4 w, u32x4& lo, u32x4& hi) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lo = (u32x4){fp8x2_to_bf16x2(w0, false), fp8x2_to_bf16x2(w0, true), fp8x2_to_bf16x2(w1, false), fp8x2_to_bf16x2(w1, true)};
    hi = (u32x4){fp8x2_to_bf16x2(w2, false), fp8x2_to_bf16x2(w2, true), fp8x2_to_bf16x2(w3, false), fp8x2_to_bf16x2(w3, true)};
}

template <bool NT>
__device__ __forceinline__ u32x4 ldw(const void* p) {
    if (NT) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// MB: activation rows (1,2,4,8); R: weight rows per wave pass (2 or 4); U: chunks (one 16-byte load per lane: 512 bf16 or
// 1024 fp8 weights of a row) in flight per row; NT: non-temporal weight loads; FP8: 1-byte e4m3 weights + per-row scale.  A "unit" is R consecutive weight rows, or for SwiGLU R/2 outputs
// (gate row g, up row g+16 of the [16 gate | 16 up] interleaved layout).
// NW: waves per workgroup.  4 (256 threads, several workgroups per CU) while the staged activations fit 64 KiB; 16 (one
// 1024-thread workgroup per CU sharing one copy of x, up to 160 KiB) for batched decode through the wide down-projection
// (4 rows x 14336 = 112 KiB).
template <int MB, int R, int U, bool NT, bool FP8, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemv_bf16_kernel(GemvParams p) {
    constexpr int NTHR = NW * 64;
    constexpr int EPL = FP8 ? 16 : 8;            // weights per lane per load
    constexpr int CH = 64 * EPL;                 // k per chunk
    constexpr int WBYTES = FP8 ? 1 : 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // xs[MB][K] bf16, then scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    float* red = (float*)(smem + (size_t)MB * K * 2);     // NW floats
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int nchunk = K / CH;
    const char* wbase = FP8 ? (const char*)p.W8 : (const char*)p.W;

    auto row_of = [&](int u, int r) -> int {
        if (swiglu) {                                  // outputs j = u*(R/2) + r/2 ; r even = gate, odd = up
            const int j = u * (R / 2) + (r >> 1);
            return (j >> 4) * 32 + (j & 15) + ((r & 1) ? 16 : 0);
        }
        const int n = u * R + r;
        return n < p.N ? n : p.N - 1;
    };

    // ---- first tile of this wave's first unit goes in flight before the prologue ----
    const int u_first = blockIdx.x * NW + wave;
    u32x4 wreg[R][U];
    if (u_first < p.units) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const char* wp = wbase + ((size_t)row_of(u_first, r) * p.ldw + lane * EPL) * WBYTES;
#pragma unroll
            for (int c = 0; c < U; ++c)
                if (c < nchunk) wreg[r][c] = ldw<NT>(wp + (size_t)c * CH * WBYTES);
        }
    }

    // ---- prologue: x (optionally RMS-normalised) -> LDS as bf16 ----
    if constexpr (MB >= 4) {
        // one wave per activation row (rows m = wave, wave + NW, ...): the sum of squares is a wave reduction, no workgroup barrier
        // per row - with 4 or 8 rows the row-after-row form below costs 4-8x the prologue of a single row
        for (int m = wave; m < MB; m += NW) {
            bf16_t* xs = (bf16_t*)smem + (size_t)m * K;
            if (m >= p.M) {
                for (int k = lane * 8; k < K; k += 64 * 8) *(uint4*)(xs + k) = make_uint4(0, 0, 0, 0);
                continue;
            }
            const bf16_t* x = p.A + (size_t)m * p.lda;
            if (p.norm_w) {
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
                    *(u16x8*)(xs + k) = o;
                }
            } else {
                for (int k = lane * 8; k < K; k += 64 * 8) *(uint4*)(xs + k) = *(const uint4*)(x + k);
            }
        }
    } else {
    for (int m = 0; m < MB; ++m) {
        bf16_t* xs = (bf16_t*)smem + (size_t)m * K;
        if (m >= p.M) {
            for (int k = tid * 8; k < K; k += NTHR * 8) *(uint4*)(xs + k) = make_uint4(0, 0, 0, 0);
            continue;
        }
        const bf16_t* x = p.A + (size_t)m * p.lda;
        if (p.norm_w) {
            float ss = 0.f;
            for (int k = tid * 8; k < K; k += NTHR * 8) {
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
            for (int w2 = 0; w2 < NW; ++w2) tot += red[w2];
            const float rstd = rsqrtf(tot / (float)K + p.norm_eps);
            for (int k = tid * 8; k < K; k += NTHR * 8) {
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
        } else {
            for (int k = tid * 8; k < K; k += NTHR * 8) *(uint4*)(xs + k) = *(const uint4*)(x + k);
        }
    }
    }
    __syncthreads();

    for (int u = u_first; u < p.units; u += gridDim.x * NW) {
        float acc[R][MB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
        const char* wp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wp[r] = wbase + ((size_t)row_of(u, r) * p.ldw + lane * EPL) * WBYTES;
        for (int c0 = 0; c0 < nchunk; c0 += U) {
            if (!(u == u_first && c0 == 0)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < U; ++c)
                        if (c0 + c < nchunk) wreg[r][c] = ldw<NT>(wp[r] + (size_t)(c0 + c) * CH * WBYTES);
            }
#pragma unroll
            for (int c = 0; c < U; ++c) {
                if (c0 + c < nchunk) {
                    if constexpr (FP8) {
                        u32x4 wl[R], wh[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) fp8x16_to_bf16(wreg[r][c], wl[r], wh[r]);
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            const bf16_t* xp = (const bf16_t*)smem + (size_t)m * K + (c0 + c) * CH + lane * 16;
                            const u32x4 x0 = *(const u32x4*)xp, x1 = *(const u32x4*)(xp + 8);
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[r][m] = dot8(wh[r], x1, dot8(wl[r], x0, acc[r][m]));
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < MB; ++m) {
                            const u32x4 xv = *(const u32x4*)((const bf16_t*)smem + (size_t)m * K + (c0 + c) * 512 + lane * 8);
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[r][m] = dot8(wreg[r][c], xv, acc[r][m]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum_lane63(acc[r][m]);     // R x MB reductions per unit: DPP, not LDS shuffles
        if (lane == 63) {
            if constexpr (FP8) {      // the row's power-of-two scale, once per output
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float sc = p.wscale[row_of(u, r)];
#pragma unroll
                    for (int m = 0; m < MB; ++m) acc[r][m] *= sc;
                }
            }
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m >= p.M) break;
                if (swiglu) {
#pragma unroll
                    for (int h = 0; h < R / 2; ++h) {
                        const int j = u * (R / 2) + h;
                        float t = act_silu(acc[2 * h][m]) * acc[2 * h + 1][m];
                        if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + j]);
                        if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + j] = t;
                        else ((bf16_t*)p.C)[(size_t)m * p.ldc + j] = f32_to_bf16(t);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int n = u * R + r;
                        if (n >= p.N) break;
                        float t = acc[r][m];
                        if (p.bias) t += p.bias[n];
                        t = apply_act(t, p.act);
                        if (p.residual) t += bf16_to_f32(p.residual[(size_t)m * p.ldr + n]);
                        if (p.out_fp32) ((float*)p.C)[(size_t)m * p.ldc + n] = t;
                        else ((bf16_t*)p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(t);
                    }
                }
            }
        }
    }
}

int g_gemv_variant = 0;   // 0 = production choice; >0 = tuning variants (tools/bench_kernels.py)

template <int MB, int R, int U, bool NT, bool FP8 = false, int NW = 4>
int launch_variant(const GemvParams& p0, hipStream_t s, size_t lds) {
    GemvParams p = p0;
    p.units = p.act == VZ_ACT_SWIGLU ? p.N / R : (p.N + R - 1) / R;     // SwiGLU: R/2 outputs of N/2 per unit
    int blocks = (p.units + NW - 1) / NW;
    int cap = NW == 4 ? 2048 : 512;
    if (g_gemv_variant >= 7 && g_gemv_variant <= 10) cap = (g_gemv_variant & 1) ? 256 : 512;      // experiments: one / two workgroups per CU
    if (blocks > cap) blocks = cap;
    static VzDeviceOnce attr;
    if (vz_device_first(attr)) {
        VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<MB, R, U, NT, FP8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, NW == 4 ? 64 * 1024 : 160 * 1024));
    }
    vz_launch_timed(gemv_bf16_kernel<MB, R, U, NT, FP8, NW>, dim3(blocks), dim3(NW * 64), lds, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

template <int MB>
int launch_mb(const GemvParams& p, hipStream_t s, size_t lds) {
    if (lds > 64 * 1024) {        // x does not fit 64 KiB: one 1024-thread workgroup per CU shares one copy (up to 160 KiB)
        if constexpr (MB >= 2) {
            if (p.W8) return launch_variant<MB, 2, 8, true, true, 16>(p, s, lds);
            return launch_variant<MB, 2, 8, true, false, 16>(p, s, lds);
        } else {
            return VZ_ERR_ARG;
        }
    }
    // Batch 1: 8-wave workgroups, at most two per CU (512), where that deals the units evenly over the CUs (O / down: 256 workgroups,
    // gate-up / lm_head: 512 looping) - measured 337 vs 331 tok/s end to end (GEMV 5.67 vs 5.56 TB/s): fewer, fatter workgroups whose
    // waves start together burst together.  QKV (3072 units = 384 such workgroups = 1.5 per CU) keeps 768 4-wave workgroups.
    bool fat = false;
    if constexpr (MB <= 4) {
        const int units = p.act == VZ_ACT_SWIGLU ? p.N / 2 : (p.N + 1) / 2;
        const int b8 = (units + 7) / 8;
        fat = g_gemv_variant == 0 && (b8 >= 512 || b8 % 256 == 0);
    }
    if (p.W8) {       // 2 rows x up to 8 chunks of 1024 k per wave (measured faster than 4 rows x 4: 13.5 vs 15.3 us on down-proj)
        if (g_gemv_variant == 1) return launch_variant<MB, 4, 4, true, true>(p, s, lds);
        if constexpr (MB <= 4) { if (fat) return launch_variant<MB, 2, 8, true, true, 8>(p, s, lds); }
        return launch_variant<MB, 2, 8, true, true>(p, s, lds);
    }
    if constexpr (MB <= 4) { if (fat) return launch_variant<MB, 2, 8, true, false, 8>(p, s, lds); }
    switch (g_gemv_variant) {
        case 1: return launch_variant<MB, 2, 4, false>(p, s, lds);
        case 2: return launch_variant<MB, 2, 8, false>(p, s, lds);
        case 3: return launch_variant<MB, 2, 8, true>(p, s, lds);
        case 4: return launch_variant<MB, 4, 4, true>(p, s, lds);
        case 5: return launch_variant<MB, 4, 4, false>(p, s, lds);
        case 6: return launch_variant<MB, 2, 4, true>(p, s, lds);
        case 7: case 8: return launch_variant<MB, 2, 8, true, false, 16>(p, s, lds);     // 16 waves, 256 / 512 workgroups
        case 9: case 10: return launch_variant<MB, 2, 8, true, false, 8>(p, s, lds);    // 8 waves, 256 / 512 workgroups
        default: return launch_variant<MB, 2, 8, true>(p, s, lds);
    }
}

}  // namespace

void vz_set_gemv_variant(int v) { g_gemv_variant = v; }

bool vz_gemv_ok(const LinearArgs& a) {
    if (a.M > 8 || (a.K % (a.W8 ? 1024 : 512)) != 0) return false;
    if (a.W8 && (!a.wscale || (a.ldw & 15) != 0 || ((uintptr_t)a.W8 & 15) != 0)) return false;
    if (a.act == VZ_ACT_SWIGLU && (a.N % 64) != 0) return false;
    const int mb = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
    return (size_t)mb * a.K * 2 + 256 <= 160 * 1024;
}

int vz_init_gemv_kernels() {
    // every variant sets its own dynamic-LDS limit on first use; make the production ones resident now so the
    // first use never happens inside a stream capture
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<1, 2, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<1, 2, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<1, 2, 8, true, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<1, 2, 8, true, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<2, 2, 8, true, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<2, 2, 8, true, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<4, 2, 8, true, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<4, 2, 8, true, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<2, 2, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<2, 2, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<4, 2, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<4, 2, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<8, 2, 8, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<8, 2, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    return VZ_OK;
}

int vz_launch_gemv(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(vz_gemv_ok(a), "gemv: needs M <= 8, K %% 512 == 0 (fp8 weights: K %% 1024 == 0, 16-byte-aligned rows, scales) and "
                                "M*K*2 <= 64 KiB (M=%d K=%d)", a.M, a.K);
    GemvParams p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual; p.norm_w = a.norm_w;
    p.W8 = a.W8; p.wscale = a.wscale;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32; p.norm_eps = a.norm_eps; p.units = 0;
    const int mb = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
    const size_t lds = (size_t)mb * a.K * 2 + 256;
    { int r = vz_init_gemv_kernels(); if (r) return r; }
    switch (mb) {
        case 1: return launch_mb<1>(p, s, lds);
        case 2: return launch_mb<2>(p, s, lds);
        case 4: return launch_mb<4>(p, s, lds);
        default: return launch_mb<8>(p, s, lds);
    }
}
