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
#include "vz_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct GemvParams {
    const bf16_t* A; const bf16_t* W; void* C;
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

template <bool NT>
__device__ __forceinline__ u32x4 ldw(const bf16_t* p) {
    if (NT) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// MB: activation rows (1,2,4,8); R: weight rows per wave pass (2 or 4); U: 512-element chunks in flight per row;
// NT: non-temporal weight loads.  A "unit" is R consecutive weight rows, or for SwiGLU R/2 outputs
// (gate row g, up row g+16 of the [16 gate | 16 up] interleaved layout).
template <int MB, int R, int U, bool NT>
__global__ __launch_bounds__(256) void gemv_bf16_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // xs[MB][K] bf16, then scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    float* red = (float*)(smem + (size_t)MB * K * 2);
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int nchunk = K >> 9;

    auto row_of = [&](int u, int r) -> int {
        if (swiglu) {                                  // outputs j = u*(R/2) + r/2 ; r even = gate, odd = up
            const int j = u * (R / 2) + (r >> 1);
            return (j >> 4) * 32 + (j & 15) + ((r & 1) ? 16 : 0);
        }
        const int n = u * R + r;
        return n < p.N ? n : p.N - 1;
    };

    // ---- first tile of this wave's first unit goes in flight before the prologue ----
    const int u_first = blockIdx.x * 4 + wave;
    u32x4 wreg[R][U];
    if (u_first < p.units) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bf16_t* wp = p.W + (size_t)row_of(u_first, r) * p.ldw + lane * 8;
#pragma unroll
            for (int c = 0; c < U; ++c)
                if (c < nchunk) wreg[r][c] = ldw<NT>(wp + c * 512);
        }
    }

    // ---- prologue: x (optionally RMS-normalised) -> LDS as bf16 ----
    for (int m = 0; m < MB; ++m) {
        bf16_t* xs = (bf16_t*)smem + (size_t)m * K;
        if (m >= p.M) {
            for (int k = tid * 8; k < K; k += 256 * 8) *(uint4*)(xs + k) = make_uint4(0, 0, 0, 0);
            continue;
        }
        const bf16_t* x = p.A + (size_t)m * p.lda;
        if (p.norm_w) {
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
            const float tot = red[0] + red[1] + red[2] + red[3];
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
        } else {
            for (int k = tid * 8; k < K; k += 256 * 8) *(uint4*)(xs + k) = *(const uint4*)(x + k);
        }
    }
    __syncthreads();

    for (int u = u_first; u < p.units; u += gridDim.x * 4) {
        float acc[R][MB];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;
        const bf16_t* wp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wp[r] = p.W + (size_t)row_of(u, r) * p.ldw + lane * 8;
        for (int c0 = 0; c0 < nchunk; c0 += U) {
            if (!(u == u_first && c0 == 0)) {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int c = 0; c < U; ++c)
                        if (c0 + c < nchunk) wreg[r][c] = ldw<NT>(wp[r] + (c0 + c) * 512);
            }
#pragma unroll
            for (int c = 0; c < U; ++c) {
                if (c0 + c < nchunk) {
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const u32x4 xv = *(const u32x4*)((const bf16_t*)smem + (size_t)m * K + (c0 + c) * 512 + lane * 8);
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r][m] = dot8(wreg[r][c], xv, acc[r][m]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);
        if (lane == 0) {
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

template <int MB, int R, int U, bool NT>
int launch_variant(const GemvParams& p0, hipStream_t s, size_t lds) {
    GemvParams p = p0;
    p.units = p.act == VZ_ACT_SWIGLU ? p.N / R : (p.N + R - 1) / R;     // SwiGLU: R/2 outputs of N/2 per unit
    int blocks = (p.units + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    static bool attr = false;
    if (!attr) {
        VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<MB, R, U, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr = true;
    }
    vz_launch_timed(gemv_bf16_kernel<MB, R, U, NT>, dim3(blocks), dim3(256), lds, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

template <int MB>
int launch_mb(const GemvParams& p, hipStream_t s, size_t lds) {
    switch (g_gemv_variant) {
        case 1: return launch_variant<MB, 2, 4, false>(p, s, lds);
        case 2: return launch_variant<MB, 2, 8, false>(p, s, lds);
        case 3: return launch_variant<MB, 2, 8, true>(p, s, lds);
        case 4: return launch_variant<MB, 4, 4, true>(p, s, lds);
        case 5: return launch_variant<MB, 4, 4, false>(p, s, lds);
        case 6: return launch_variant<MB, 2, 4, true>(p, s, lds);
        default: return launch_variant<MB, 2, 8, true>(p, s, lds);
    }
}

}  // namespace

void vz_set_gemv_variant(int v) { g_gemv_variant = v; }

bool vz_gemv_ok(const LinearArgs& a) {
    if (a.M > 8 || (a.K % 512) != 0) return false;
    if (a.act == VZ_ACT_SWIGLU && (a.N % 64) != 0) return false;
    const int mb = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
    return (size_t)mb * a.K * 2 + 64 <= 64 * 1024;
}

int vz_init_gemv_kernels() {
    // every variant sets its own dynamic-LDS limit on first use; make the production ones resident now so the
    // first use never happens inside a stream capture
    static bool done = false;
    if (done) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<1, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<2, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<4, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemv_bf16_kernel<8, 2, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    done = true;
    return VZ_OK;
}

int vz_launch_gemv(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(vz_gemv_ok(a), "gemv: needs M <= 8, K %% 512 == 0 and M*K*2 <= 64 KiB (M=%d K=%d)", a.M, a.K);
    GemvParams p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual; p.norm_w = a.norm_w;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32; p.norm_eps = a.norm_eps; p.units = 0;
    const int mb = a.M <= 1 ? 1 : a.M <= 2 ? 2 : a.M <= 4 ? 4 : 8;
    const size_t lds = (size_t)mb * a.K * 2 + 64;
    { int r = vz_init_gemv_kernels(); if (r) return r; }
    switch (mb) {
        case 1: return launch_mb<1>(p, s, lds);
        case 2: return launch_mb<2>(p, s, lds);
        case 4: return launch_mb<4>(p, s, lds);
        default: return launch_mb<8>(p, s, lds);
    }
}
