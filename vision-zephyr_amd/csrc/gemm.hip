// Linear layers of the Vision-Zephyr hot path on gfx950:  C[M,N'] = epi(A[M,K] . W[N,K]^T)
//
//   * gemm_bf16_kernel : 128x128x64 MFMA tile GEMM (v_mfma_f32_16x16x32_bf16), operands staged
//     global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4) into a lane-linear image whose
//     XOR swizzle is applied on the SOURCE address and again on the ds_read_b128 (conflict-free
//     for the 16-row fragment reads), double-buffered, XCD-aware tile order.  Used for every
//     M > 8 product: CLIP QKV/out/MLP, Q-Former projections, Zephyr prefill QKV/O/gate-up/down,
//     lm_head over all positions.
//   * gemv_bf16_kernel : weight-streaming GEMV for M <= 8 (decode): each wave streams two weight
//     rows with 16-byte loads straight to VGPRs (no LDS round trip for data read once), x staged
//     once per block in LDS, optional fused RMSNorm prologue, v_dot2c_f32_bf16 accumulate.
//
// Both take W as the reference stores it ([out_features, in_features], K contiguous), so no
// transposed copy of the 7.2 B parameters is ever made.  Epilogues (bias, quick_gelu, erf-GELU,
// SwiGLU pair, residual add, bf16/fp32 out) run on the fp32 accumulators before the single
// rounding to bf16 - the rounding points the bf16 oracle mirrors (oracle/vz_oracle.py).
//
// Semantics: hf:models/clip/modeling_clip.py:280-350, hf:models/mistral/modeling_mistral.py:35-48,
// 122-178,450-453, torch.nn.MultiheadAttention projections (ref:vis_zephyr/model/multimodal_projector/builder.py:16-32).
#include <algorithm>

#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>

#include "vz_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;      // A + W
constexpr int GEMM_LDS = 2 * BUF_BYTES;        // double buffer: 64 KiB -> 2 workgroups / CU

struct GemmParams {
    const bf16_t* A; const bf16_t* W; void* C;
    const float* bias; const bf16_t* residual;
    int M, N, K, lda, ldw, ldc, ldr;
    int act, out_fp32, tiles_m, tiles_n;
    int splitk; float* slab;   // splitk > 1: fp32 partial tiles go to slab[split][M][N], epilogue runs in splitk_finalize_kernel
    // batched launches (blockIdx.y = o * n_inner + i; the training step's attention contractions over (sample, head)): element
    // offsets o * so + (i / div) * si per operand; n_inner == 0: a single problem
    int n_inner; int a_div, w_div;
    long a_so, a_si, w_so, w_si, c_so, c_si;
};

__device__ __forceinline__ void glds16(const char* g, char* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware); source address is per lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// ---- fast epilogue: all 128 columns of the tile inside N, 8-byte-aligned output rows, 16-byte-aligned bias.  Everything uniform (activation, bias /
// residual present, output type) is decided once outside the per-element code; the generic path below spends ~10x longer
// in divergent per-element branches (measured with in-kernel stamps on the 256x256 kernel: 29 us vs 5 us per tile).
template <int ACT>
__device__ __forceinline__ f32x4 act4(f32x4 v) {
    if constexpr (ACT == VZ_ACT_QUICK_GELU) return (f32x4){act_quick_gelu(v[0]), act_quick_gelu(v[1]), act_quick_gelu(v[2]), act_quick_gelu(v[3])};
    else if constexpr (ACT == VZ_ACT_GELU_ERF) return (f32x4){act_gelu_erf(v[0]), act_gelu_erf(v[1]), act_gelu_erf(v[2]), act_gelu_erf(v[3])};
    else return v;
}

__device__ __forceinline__ void put4(const GemmParams& p, bool has_res, bool f32, int m, int n0, f32x4 v) {
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

template <int ACT, int MT>
__device__ __forceinline__ void epilogue_fast_rows(const GemmParams& p, f32x4 (&acc)[4][4], const f32x4 (&b4)[4], int m_base, int n_base,
                                                   int n_half, bool has_res, bool f32) {
    const int m = m_base + MT * 16;
    if (m >= p.M) return;                   // the last tile row may be partial; columns never are on this path
    if constexpr (ACT == VZ_ACT_SWIGLU) {   // weight rows interleaved [16 gate | 16 up]: tile nt = gate, nt+1 = up, same lane slots
#pragma unroll
        for (int nt = 0; nt < 4; nt += 2) {
            const f32x4 gt = acc[nt][MT], up = acc[nt + 1][MT];
            put4(p, has_res, f32, m, n_half + (nt >> 1) * 16,
                 (f32x4){act_silu(gt[0]) * up[0], act_silu(gt[1]) * up[1], act_silu(gt[2]) * up[2], act_silu(gt[3]) * up[3]});
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) put4(p, has_res, f32, m, n_base + nt * 16, act4<ACT>(acc[nt][MT] + b4[nt]));
    }
}

template <int ACT>
__device__ __forceinline__ void epilogue_fast(const GemmParams& p, f32x4 (&acc)[4][4], int m_base, int n_base, int n_half) {
    const bool has_res = p.residual != nullptr, f32 = p.out_fp32 != 0;
    f32x4 b4[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
        b4[nt] = (ACT != VZ_ACT_SWIGLU && p.bias) ? *(const f32x4*)(p.bias + n_base + nt * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
    epilogue_fast_rows<ACT, 0>(p, acc, b4, m_base, n_base, n_half, has_res, f32);
    epilogue_fast_rows<ACT, 1>(p, acc, b4, m_base, n_base, n_half, has_res, f32);
    epilogue_fast_rows<ACT, 2>(p, acc, b4, m_base, n_base, n_half, has_res, f32);
    epilogue_fast_rows<ACT, 3>(p, acc, b4, m_base, n_base, n_half, has_res, f32);
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.n_inner > 0) {
        const int o = blockIdx.y / p.n_inner, i = blockIdx.y - o * p.n_inner;
        p.A += o * p.a_so + (i / p.a_div) * p.a_si;
        p.W += o * p.w_so + (i / p.w_div) * p.w_si;
        p.C = p.out_fp32 ? (void*)((float*)p.C + o * p.c_so + i * p.c_si) : (void*)((bf16_t*)p.C + o * p.c_so + i * p.c_si);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, 64(m) x 64(n) each

    // XCD-aware, bijective tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
    // contiguous run of tiles; inside a run tiles walk M first, i.e. neighbours share one W panel.
    const int nwg = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nwg;              // split-K slices of one tile are launched nwg blocks apart
    const int bid = blockIdx.x - split * nwg;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bn = tile / p.tiles_m, bm = tile - bn * p.tiles_m;

    // ---- staging addresses: 1024 16-byte chunks per operand tile, 4 per thread ----
    const char* ga[4];
    const char* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = i * 256 + tid;
        const int row = ch >> 3, c = ch & 7;
        const int gc = c ^ (row & 7);  // swizzle on the source; the LDS image stays lane-linear
        int arow = bm * BM + row; arow = arow < p.M ? arow : p.M - 1;
        int wrow = bn * BN + row; wrow = wrow < p.N ? wrow : p.N - 1;
        ga[i] = (const char*)p.A + ((size_t)arow * p.lda) * 2 + gc * 16;
        gw[i] = (const char*)p.W + ((size_t)wrow * p.ldw) * 2 + gc * 16;
    }
    const int wave_chunk = wave * 64 * 16;

    // ---- fragment read addresses (16x16x32: lane holds row lane&15, k = 8*(lane>>4)+j) ----
    const int frow = lane & 15, g = lane >> 4;
    const int koff0 = ((g ^ (lane & 7)) << 4);  // k-step 0; k-step 1 = koff0 ^ 64
    const int a_off = (wm * 64 + frow) * 128;
    const int w_off = TILE_BYTES + (wn * 64 + frow) * 128;

    f32x4 acc[4][4];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk_all = p.K / BK;
    const int k_lo = (int)((long)nk_all * split / p.splitk), k_hi = (int)((long)nk_all * (split + 1) / p.splitk);
    const int nk = k_hi - k_lo;
    // A rows past M are never stored: their 32-row staging slices are not loaded at all (a 64-row decode step moves half the activation
    // bytes per tile; tools/micro/stream_bench.hip: activation traffic through the vector-memory path costs the weight stream 20-34 %).
    // The LDS rows keep whatever they held; the accumulators of those rows are dead.
    const int a_slices = min(4, (p.M - bm * BM + 31) >> 5);
    auto stage = [&](int buf, int kt_rel) {
        const int kt = k_lo + kt_rel;
        char* la = smem + buf * BUF_BYTES + wave_chunk;
        char* lw = la + TILE_BYTES;
        const int kb = kt * (BK * 2);
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
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* base = smem + cur * BUF_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ko = koff0 ^ (ks * 64);
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(base + a_off + t * 2048 + ko);
                wf[t] = *(const bf16x8*)(base + w_off + t * 2048 + ko);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: acc[nt][mt][j] = C[m = .. + mt*16 + (lane&15)][n = .. + nt*16 + 4*(lane>>4) + j] ----
    const int m_base = bm * BM + wm * 64 + frow;
    const int n_base = bn * BN + wn * 64 + g * 4;
    if (p.splitk > 1) {   // raw fp32 partial sums; bias / activation / residual / rounding happen once, in the finalize kernel
        float* slab = p.slab + (size_t)split * p.M * p.N;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = m_base + mt * 16;
            if (m >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n0 = n_base + nt * 16;
                if (n0 + 3 < p.N && (p.N & 3) == 0) *(f32x4*)(slab + (size_t)m * p.N + n0) = acc[nt][mt];
                else
                    for (int j = 0; j < 4; ++j)
                        if (n0 + j < p.N) slab[(size_t)m * p.N + n0 + j] = acc[nt][mt][j];
            }
        }
        return;
    }
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int n_out_total = swiglu ? p.N / 2 : p.N;
    const bool vec_ok = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0);
    if (bn * BN + BN <= p.N && vec_ok && (((uintptr_t)p.bias) & 15) == 0) {
        const int n_half = ((bn * BN + wn * 64) >> 1) + g * 4;
        switch (p.act) {
            case VZ_ACT_QUICK_GELU: epilogue_fast<VZ_ACT_QUICK_GELU>(p, acc, m_base, n_base, n_half); break;
            case VZ_ACT_GELU_ERF: epilogue_fast<VZ_ACT_GELU_ERF>(p, acc, m_base, n_base, n_half); break;
            case VZ_ACT_SWIGLU: epilogue_fast<VZ_ACT_SWIGLU>(p, acc, m_base, n_base, n_half); break;
            default: epilogue_fast<VZ_ACT_NONE>(p, acc, m_base, n_base, n_half); break;
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m_base + mt * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            float v[4];
            int n0;
            if (swiglu) {
                if (nt & 1) continue;
                // weight rows are interleaved [16 gate | 16 up]: tile nt = gate, nt+1 = up, same lane slots
                n0 = ((bn * BN + wn * 64) >> 1) + (nt >> 1) * 16 + g * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_silu(acc[nt][mt][j]) * acc[nt + 1][mt][j];
            } else {
                n0 = n_base + nt * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[nt][mt][j];
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

// sum of the split-K slabs + the fused epilogue (bias / activation / residual / bf16|fp32 out), 4 outputs per thread.
// SwiGLU (decode batches of 33..64 rows cut the gate|up GEMM in two along K): the slabs hold the raw [M, N = 2I] sums in the
// interleaved [16 gate | 16 up] column order; output column c of [M, I] pairs slab columns (c >> 4) * 32 + (c & 15) and + 16.
// Slices are requested eight at a time and added in split order; bias, residual and the output go as one vector access per quad when the
// leading dimensions allow (the first form - one load and a full wait per slice, per-element bias / residual loads and 2-byte stores -
// took 7-8 us per call: 22 us of a 64-row decode layer, 0.6 ms of the first token).  Same fp32 order: bit-identical.
__device__ __forceinline__ f32x4 slab_sum(const float* base, size_t sstride, int splitk) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < splitk; s0 += 8) {
        f32x4 t[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) t[d] = *(const f32x4*)(base + (size_t)min(s0 + d, splitk - 1) * sstride);
#pragma unroll
        for (int d = 0; d < 8; ++d)
            if (s0 + d < splitk) v += t[d];
    }
    return v;
}

__global__ __launch_bounds__(256) void splitk_finalize_kernel(GemmParams p) {
    const bool swiglu = p.act == VZ_ACT_SWIGLU;
    const int n_out = swiglu ? p.N / 2 : p.N;
    const long quads = (long)p.M * (n_out / 4);
    const size_t sstride = (size_t)p.M * p.N;
    const bool vec = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) && (((uintptr_t)p.bias) & 15) == 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < quads; i += (long)gridDim.x * 256) {
        const int m = (int)(i / (n_out / 4)), n0 = (int)(i % (n_out / 4)) * 4;
        f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
        if (p.residual) {
            if (vec) {
                const u16x4 rr = *(const u16x4*)(p.residual + (size_t)m * p.ldr + n0);
                r4 = (f32x4){bf16_to_f32(rr[0]), bf16_to_f32(rr[1]), bf16_to_f32(rr[2]), bf16_to_f32(rr[3])};
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) r4[j] = bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + j]);
            }
        }
        float o[4];
        if (swiglu) {
            const int gc = (n0 >> 4) * 32 + (n0 & 15);
            const float* row = p.slab + (size_t)m * p.N;
            const f32x4 g = slab_sum(row + gc, sstride, p.splitk), u = slab_sum(row + gc + 16, sstride, p.splitk);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = act_silu(g[j]) * u[j];
                if (p.residual) t += r4[j];
                o[j] = t;
            }
        } else {
            const f32x4 v = slab_sum(p.slab + (size_t)m * p.N + n0, sstride, p.splitk);
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
                if (vec) b4 = *(const f32x4*)(p.bias + n0);
                else { b4[0] = p.bias[n0]; b4[1] = p.bias[n0 + 1]; b4[2] = p.bias[n0 + 2]; b4[3] = p.bias[n0 + 3]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j];
                if (p.bias) t += b4[j];
                t = apply_act(t, p.act);
                if (p.residual) t += r4[j];
                o[j] = t;
            }
        }
        if (p.out_fp32) {
            float* c = (float*)p.C + (size_t)m * p.ldc + n0;
            if (vec) *(f32x4*)c = (f32x4){o[0], o[1], o[2], o[3]};
            else { c[0] = o[0]; c[1] = o[1]; c[2] = o[2]; c[3] = o[3]; }
        } else {
            bf16_t* c = (bf16_t*)p.C + (size_t)m * p.ldc + n0;
            if (vec) {
                uint2 pk;
                pk.x = pack_bf16x2(o[0], o[1]);
                pk.y = pack_bf16x2(o[2], o[3]);
                *(uint2*)c = pk;
            } else { c[0] = f32_to_bf16(o[0]); c[1] = f32_to_bf16(o[1]); c[2] = f32_to_bf16(o[2]); c[3] = f32_to_bf16(o[3]); }
        }
    }
}

}  // namespace

int vz_linear_check_common(const LinearArgs& a) {
    VZ_CHECK_ARG(a.A && a.W && a.C, "linear: null pointer");
    VZ_CHECK_ARG(a.M > 0 && a.N > 0 && a.K > 0, "linear: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    VZ_CHECK_ARG(a.K % 64 == 0, "linear: K=%d must be a multiple of 64", a.K);
    VZ_CHECK_ARG(a.lda >= a.K && a.ldw >= a.K && (a.lda % 8) == 0 && (a.ldw % 8) == 0,
                 "linear: lda=%d ldw=%d must be >= K and multiples of 8", a.lda, a.ldw);
    VZ_CHECK_ARG(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && ((uintptr_t)a.C & 15) == 0,
                 "linear: pointers must be 16-byte aligned");
    VZ_CHECK_ARG(a.act >= 0 && a.act <= 3, "linear: unknown activation %d", a.act);
    if (a.act == VZ_ACT_SWIGLU) {
        VZ_CHECK_ARG(a.N % 32 == 0 && !a.bias, "linear: SwiGLU needs N %% 32 == 0 and no bias");
        VZ_CHECK_ARG(a.ldc >= a.N / 2, "linear: ldc too small");
    } else {
        VZ_CHECK_ARG(a.ldc >= a.N, "linear: ldc=%d < N=%d", a.ldc, a.N);
    }
    VZ_CHECK_ARG(!a.residual || ((uintptr_t)a.residual & 7) == 0, "linear: residual must be 8-byte aligned");
    return VZ_OK;
}

// ---- per-(device, stream) scratch ----
// Split-K slabs and arrival tickets are written by one launch and read by the next: launches that share them must be ordered.  They
// therefore exist per (kind, device, stream) - two engines, two streams or two devices in one process (the reference's API server
// runs generate() in a thread per request, ref:vis_zephyr/serve/api.py:148-184) never meet on one slab - like the stream-K state of
// gemm256.hip.  Nothing is allocated inside a stream capture: a capturing launcher that finds no scratch gets nullptr and takes its
// whole-K route; vz_llm_decode_steps reserves the scratch of its capture stream before it captures.
namespace {
struct StreamWs { void* p = nullptr; size_t bytes = 0; };
std::mutex g_ws_mu;
std::map<std::tuple<int, int, hipStream_t>, StreamWs> g_ws;
}  // namespace

int vz_stream_ws(int kind, hipStream_t s, size_t min_bytes, bool zero, void** out, size_t* out_bytes) {
    int dev = 0;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ws_mu);
    StreamWs& w = g_ws[std::make_tuple(kind, dev, s)];
    if (w.bytes < min_bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) {
            if (w.p) { VZ_CHECK_HIP(hipDeviceSynchronize()); VZ_CHECK_HIP(hipFree(w.p)); w.p = nullptr; w.bytes = 0; }
            VZ_CHECK_HIP(hipMalloc(&w.p, min_bytes));
            if (zero) VZ_CHECK_HIP(hipMemset(w.p, 0, min_bytes));
            w.bytes = min_bytes;
        }
    }
    *out = w.p;                       // (inside a capture: whatever exists, possibly nothing / too small - the caller checks *out_bytes)
    if (out_bytes) *out_bytes = w.bytes;
    return VZ_OK;
}

bool vz_device_first(VzDeviceOnce& o) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) dev = 0;
    std::lock_guard<std::mutex> lk(mu);
    const unsigned long long bit = 1ull << (dev & 63);
    if (o.seen[dev >> 6] & bit) return false;
    o.seen[dev >> 6] |= bit;
    return true;
}

constexpr size_t SLAB_DEFAULT = (size_t)96 << 20;    // every split-K shape a captured decode step reaches (64 rows x 8 slices of the lm_head) and the Q-Former's (M <= 512)
int vz_init_gemm_kernels() {
    static VzDeviceOnce once;                        // per device: hipFuncSetAttribute is a per-device setting
    if (!vz_device_first(once)) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)gemm_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS));
    { int r = vz_init_gemv_kernels(); if (r) return r; r = vz_init_gemm256_kernel(); if (r) return r; r = vz_init_skinny_kernels(); if (r) return r; r = vz_init_wide_kernels(); if (r) return r; r = vz_init_sampling_kernels(); if (r) return r; }
    return VZ_OK;
}

static int g_gemm_choice = 0;
void vz_set_gemm_choice(int v) { g_gemm_choice = v; }
static int g_splitk_mode = 0;          // 0 auto, 1 never (A/B knob 3)
void vz_set_splitk_mode(int v) { g_splitk_mode = v; }
static int g_splitk_mid = 1;           // 1: K slices for grids of fewer than 256 tiles with M > 512 (A/B knob 26)
void vz_set_splitk_mid(int v) { g_splitk_mid = v; }
static int g_splitk_cap = 8;           // most K slices of a weight-streaming (M <= 512) product (A/B knob 24; 4 -> 8: Q-Former 3.87 -> 3.80 ms)
void vz_set_splitk_cap(int v) { g_splitk_cap = v < 1 ? 1 : (v > 16 ? 16 : v); }

// Tile choice: the 256x256 8-phase kernel runs one workgroup per CU, so it needs enough 256^2 tiles to fill the
// 256 CUs several times over (>= 512 tiles: measured cross-over on MI355X, tools/bench_kernels.py); smaller grids keep the
// 128x128 kernel (2 workgroups per CU, 4x the tiles).
int vz_launch_gemm(const LinearArgs& a, hipStream_t s) {
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    // measured on MI355X (tools/bench_kernels.py gemm): the 256^2 kernel wins once it has >= 160 tiles (QKV 192: 100 vs 118 us,
    // gate-up 896: 373 vs 506 us) or, with its stream-K tail, >= 96 tiles of long K (down-proj 128 tiles x 224 K-tiles:
    // 190 vs 252 us); the 128^2 kernel keeps short K (CLIP, K = 1024) and small grids (O-proj 128 tiles x 64: 68 vs 80 us)
    const bool use256 = g_gemm_choice == 2 || (g_gemm_choice == 0 && a.K >= 1024 && (t256 >= 160 || (t256 >= 96 && a.K >= 8192)));
    return use256 ? vz_launch_gemm256(a, s) : vz_launch_gemm128(a, s);
}

int vz_launch_gemm128(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(!a.norm_w, "linear: fused RMSNorm prologue exists on the GEMV path only");
    GemmParams p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32;
    p.tiles_m = (a.M + BM - 1) / BM;
    p.tiles_n = (a.N + BN - 1) / BN;
    { int r = vz_init_gemm_kernels(); if (r) return r; }
    // Split-K for weight-streaming shapes (M <= 512: Q-Former projections at 32 rows per tile): with a handful of row tiles
    // the grid cannot fill 256 CUs, so K is cut into up to 4 slices per tile; the fp32 slabs cost 2 x 4 x M x N x splitk bytes
    // of extra traffic, small next to the weights only while M is small.
    const int tiles = p.tiles_m * p.tiles_n, nk = a.K / BK;
    int splitk = 1;
    // The factor depends on N and K only, never on M: a row's result must not change with the number of rows beside it
    // (the Q-Former computes block 0's self-attention once per sample and relies on it being bit-identical to the
    // per-tile computation, tests/test_stages_gpu.py::test_qformer).
    (void)tiles;
    if (g_splitk_mode != 1 && a.M <= 512 && p.tiles_n < 128 && a.act != VZ_ACT_SWIGLU && (a.N & 3) == 0) {
        // ~256 workgroups per row tile, rounded DOWN: two row tiles (the Q-Former's 160 rows) then fit the 512 slots in one round
        // (N = 12288: 96 column tiles x 3 slices x 2 row tiles = 576 workgroups ran a second round - 57 us; x 2 slices: 47 us)
        splitk = std::max(1, 256 / p.tiles_n);
        if (splitk > g_splitk_cap) splitk = g_splitk_cap;
        while (splitk > 1 && nk / splitk < 8) --splitk;
    }
    // A grid that leaves the CUs a single workgroup each (or none) is cut along K until ~512 workgroups exist: a lone workgroup has nobody to
    // hide its LDS / barrier latency behind (CLIP fc2 at 5 tiles: 184 tiles x K = 4096, 50 -> 31 us; CLIP tower 4.99 -> 4.54 ms at 5 tiles,
    // 3.78 -> 2.8 at 1; tools/bench_vision.py).  Short K (1024: CLIP QKV / out) only pays on very small grids.  The factor follows the tile
    // count, i.e. M: like the 128^2 / 256^2 choice it is not batch-invariant (knob 26 = 0 switches it off; the invariance tests do).
    if (g_splitk_mid && splitk == 1 && a.M > 512 && a.act != VZ_ACT_SWIGLU && (a.N & 3) == 0 &&
        ((tiles < 256 && nk >= 32) || (tiles <= 128 && nk >= 16))) {
        splitk = std::min(4, 512 / tiles);
        while (splitk > 1 && nk / splitk < 8) --splitk;
    }
    if (a.splitk_hint > 0 && (a.N & 7) == 0) {       // decode batches (see LinearArgs.splitk_hint); SwiGLU pairs are formed in the finalize kernel
        splitk = a.splitk_hint;
        while (splitk > 1 && nk / splitk < 4) --splitk;
    }
    p.splitk = splitk; p.slab = nullptr; p.n_inner = 0; p.a_div = p.w_div = 1; p.a_so = p.a_si = p.w_so = p.w_si = p.c_so = p.c_si = 0;
    if (splitk > 1) {
        // this stream's slab (96 MiB by default, grown on demand - never inside a capture: there the factor shrinks to what the slab
        // reserved before the capture holds, down to whole-K tiles)
        size_t need = (size_t)splitk * a.M * a.N * sizeof(float), have = 0;
        void* slab = nullptr;
        { int r = vz_stream_ws(0, s, std::max(need, SLAB_DEFAULT), false, &slab, &have); if (r) return r; }
        while (splitk > 1 && (size_t)splitk * a.M * a.N * sizeof(float) > have) --splitk;
        p.splitk = splitk;
        p.slab = splitk > 1 ? (float*)slab : nullptr;
    }
    vz_launch_timed(gemm_bf16_kernel, dim3(tiles * splitk), dim3(256), GEMM_LDS, s, p);
    VZ_LAUNCH_CHECK();
    if (splitk > 1) {
        long blocks = ((long)a.M * ((a.act == VZ_ACT_SWIGLU ? a.N / 2 : a.N) / 4) + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_finalize_kernel, dim3((int)blocks), dim3(256), 0, s, p);
        VZ_LAUNCH_CHECK();
    }
    return VZ_OK;
}

// n_outer x n_inner independent products C_b = A_b . W_b^T on the 128^2 kernel (no bias / residual / activation, no split-K):
// operand b = (o, i) starts at base + o * so + (i / div) * si elements - `div` lets the 4 query heads of a KV head share one W
int vz_launch_gemm_batched(const BatchedGemmArgs& b, hipStream_t s) {
    LinearArgs a;
    a.A = b.A; a.lda = b.lda; a.W = b.W; a.ldw = b.ldw; a.C = b.C; a.ldc = b.ldc; a.M = b.M; a.N = b.N; a.K = b.K;
    a.bias = nullptr; a.residual = nullptr; a.ldr = 0; a.act = VZ_ACT_NONE; a.out_fp32 = b.out_fp32; a.norm_w = nullptr; a.norm_eps = 0.f;
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    VZ_CHECK_ARG(b.n_outer >= 1 && b.n_inner >= 1 && (long)b.n_outer * b.n_inner <= 65535 && b.a_div >= 1 && b.w_div >= 1, "gemm_batched: bad batch");
    VZ_CHECK_ARG((b.a_so % 8) == 0 && (b.a_si % 8) == 0 && (b.w_so % 8) == 0 && (b.w_si % 8) == 0 && (b.c_so % 4) == 0 && (b.c_si % 4) == 0,
                 "gemm_batched: batch strides must keep 16-byte alignment");
    { int r = vz_init_gemm_kernels(); if (r) return r; }
    GemmParams p;
    p.A = a.A; p.W = a.W; p.C = a.C; p.bias = nullptr; p.residual = nullptr;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldc = a.ldc; p.ldr = 0;
    p.act = VZ_ACT_NONE; p.out_fp32 = a.out_fp32;
    p.tiles_m = (a.M + BM - 1) / BM; p.tiles_n = (a.N + BN - 1) / BN;
    p.splitk = 1; p.slab = nullptr;
    p.n_inner = b.n_inner; p.a_div = b.a_div; p.w_div = b.w_div;
    p.a_so = b.a_so; p.a_si = b.a_si; p.w_so = b.w_so; p.w_si = b.w_si; p.c_so = b.c_so; p.c_si = b.c_si;
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3(p.tiles_m * p.tiles_n, b.n_outer * b.n_inner), dim3(256), GEMM_LDS, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_linear(const LinearArgs& a, hipStream_t s) {
    // 2 rows: the GEMV with both rows in LDS streams the weights like the 1-row launch (3.12 vs 3.48 ms per 2-row step); 3..64 rows
    // (batched decode): one MFMA per KiB of weights
    if (g_skinny_mode && vz_skinny_ok(a) && !(a.M == 2 && g_skinny_mode != 7 && vz_gemv_ok(a))) return vz_launch_skinny(a, s);
    if (vz_gemv_ok(a)) return vz_launch_gemv(a, s);
    VZ_CHECK_ARG(!a.W8, "linear: e4m3 weights are streamed by the M <= 16 kernels only (M=%d K=%d)", a.M, a.K);
    return vz_launch_gemm(a, s);
}
