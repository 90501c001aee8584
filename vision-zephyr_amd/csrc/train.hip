// Backward kernels of the Stage-1 pretrain step (ref:vis_zephyr/train/train.py:817-829 trains the
// Q-Former projector only, through the frozen Zephyr, on HF's causal-LM cross-entropy; ref:script/pretrain.sh:39-42 AdamW).
// Restated on the CPU in oracle/train_oracle.py (autograd through the pinned forward restatement).
//
// Every contraction of the backward runs on the forward's MFMA tile GEMMs (gemm.hip / gemm256.hip compute C = A . W^T with both
// operands K-contiguous):
//     input gradient   dX[R,K] = dY[R,N] . W[N,K]        ->  A = dY,   "W" = W^T [K,N]   (frozen Zephyr weights: transposed once at
//                                                             set-up; projector weights: transposed every step, 3.4 GB)
//     weight gradient  dW[N,K] = dY^T[N,R] . X[R,K]      ->  A = dY^T, "W" = X^T [K,R]   (rows R are the contraction; fp32 out)
//     attention        S = Q K^T, dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO as BATCHED tile GEMMs over (sample, head) with the
//                      probabilities materialised (Stage-1 sequences are ~200 tokens: P is 160 MB per layer), 4 query heads of a
//                      KV head concatenated along the contraction for dK / dV (the GQA sum comes out of the GEMM)
// so what lives here is the data movement around them (zero-padded batched transposes) and the row-wise / element-wise
// derivatives: masked softmax forward + backward, RMSNorm / LayerNorm backward (dx, and per-workgroup partial sums of dw / db that a
// second kernel adds in a fixed order: no float atomics, gradients are reproducible bit for bit), SwiGLU / exact-GELU backward,
// RoPE backward + re-assembly of the fused-QKV gradient, cross-entropy (loss + dlogits in one pass pair), column sums for the bias
// gradients, the fused AdamW update.  All HBM-bound: 16-byte accesses where the layout allows, one wave or one workgroup per row.
#include "vz_common.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* red) {     // 256 threads = 4 waves; result in every thread
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// dst[o][i][c][col0 + r] = src[o][i][r][c]   (bf16; 32 x 32 tiles through LDS; the caller zero-fills dst's padding)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, long src_rs, long src_so, long src_si,
                                                        bf16_t* __restrict__ dst, long dst_rs, long dst_so, long dst_si, int R, int C,
                                                        int n_inner, int col0) {
    __shared__ bf16_t tile[32][33];
    const int batch = blockIdx.z, o = batch / n_inner, i = batch - o * n_inner;
    const bf16_t* s = src + o * src_so + i * src_si;
    bf16_t* d = dst + o * dst_so + i * dst_si;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, c = c0 + tx;
        tile[ty + k * 8][tx] = (r < R && c < C) ? s[(size_t)r * src_rs + c] : (bf16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + k * 8, r = r0 + tx;
        if (c < C && r < R) d[(size_t)c * dst_rs + col0 + r] = tile[tx][ty + k * 8];
    }
}

// The same transposition in 64 x 64 tiles with 16-byte global accesses on both sides (the 32 x 32 form above moves 2 bytes per lane:
// 8.5 % of the Stage-1 step).  A thread loads two 8-element row pieces and stores them AS THEY ARE into a row-major LDS tile (16-byte
// stores, pitch 144 bytes); the transposition is the hardware's: ds_read_b64_tr_b16 hands lane c of a 16-lane group column c of a 4-row x
// 16-column block, two of them are 8 consecutive source rows of one column = one 16-byte piece of an output row.  (Round 2 scattered the
// loaded pieces transposed into LDS with sixteen 2-byte stores per thread, four lanes to a bank: 1.8 TB/s; this form: see DESIGN section 4.)
// Needs R, C, col0 and every stride a multiple of 8 elements and 16-byte-aligned bases (the launcher checks; else the 32 x 32 kernel).
template <int T>       // T x T tiles (64 or 128): a tile row is T * 2 contiguous bytes on both sides - 256-byte segments at T = 128
__global__ __launch_bounds__(256) void transpose64_kernel(const bf16_t* __restrict__ src, long src_rs, long src_so, long src_si,
                                                          bf16_t* __restrict__ dst, long dst_rs, long dst_so, long dst_si, int R, int C,
                                                          int n_inner, int col0) {
    constexpr int PITCH = T * 2 + 16, PPR = T / 8, NP = T * PPR / 256;       // pieces per tile row, 16-byte pieces per thread
    __shared__ __attribute__((aligned(16))) char tile[T * PITCH];
    const int batch = blockIdx.z, o = batch / n_inner, i = batch - o * n_inner;
    const bf16_t* s = src + o * src_so + i * src_si;
    bf16_t* d = dst + o * dst_so + i * dst_si;
    const int r0 = blockIdx.y * T, c0 = blockIdx.x * T;
    u16x8 v[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = threadIdx.x + k * 256, r = r0 + q / PPR, c = c0 + (q % PPR) * 8;
        v[k] = (r < R && c < C) ? *(const u16x8*)(s + (size_t)r * src_rs + c) : (u16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int q = threadIdx.x + k * 256;
        *(u16x8*)(tile + (q / PPR) * PITCH + (q % PPR) * 16) = v[k];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        // (column block of 16, piece of 8 source rows) of this 16-lane group: the four groups of an instruction take four neighbouring pieces
        const int combo = (wave * NP + k) * 4 + g, cb = combo / PPR, r8 = combo % PPR;
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const char* tp = tile + (r8 * 8 + (cc >> 2)) * PITCH + (cb * 16 + (cc & 3) * 4) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)tp);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tp + 4 * PITCH));
        const int c = c0 + cb * 16 + cc, r = r0 + r8 * 8;
        if (c < C && r < R) {
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            *(s16x8*)(d + (size_t)c * dst_rs + col0 + r) = (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// masked softmax over fp32 score rows: P[row][j] = softmax_j(scale * S[row][j]) over the keys the query may see, 0 elsewhere
// (incl. the padding columns up to ldp).  One wave per row.  row = ((b * H + h) * Sq + i); query i of batch b sits at position i.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, int lds_, bf16_t* __restrict__ P, int ldp, long rows,
                                                          int H, int Sq, int Sk, float scale, int causal, int window,
                                                          const int* __restrict__ kv_len) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int i = (int)(row % Sq), b = (int)(row / ((long)Sq * H));
    int hi = kv_len ? kv_len[b] : Sk;                 // keys [lo, hi) are visible
    hi = hi < Sk ? hi : Sk;
    int lo = 0;
    if (causal) { hi = hi < i + 1 ? hi : i + 1; if (window > 0 && i + 1 - window > 0) lo = i + 1 - window; }
    const float* s = S + row * lds_;
    bf16_t* p = P + row * ldp;
    float m = -INFINITY;
    for (int j = lo + lane; j < hi; j += 64) m = fmaxf(m, s[j] * scale);
    m = wave_max(m);
    float l = 0.f;
    for (int j = lo + lane; j < hi; j += 64) l += __expf(s[j] * scale - m);
    l = wave_sum(l);
    const float inv = hi > lo ? 1.f / l : 0.f;
    for (int j = lane; j < ldp; j += 64) p[j] = (j >= lo && j < hi) ? f32_to_bf16(__expf(s[j] * scale - m) * inv) : (bf16_t)0;
}

// dS = P o (dP - sum_j P dP) * scale   (bf16 out, zero where P is zero)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, int ldp, const float* __restrict__ dP, int lddp,
                                                          bf16_t* __restrict__ dS, int ldds, long rows, int Sk, float scale) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* p = P + row * ldp;
    const float* dp = dP + row * lddp;
    bf16_t* ds = dS + row * ldds;
    // (a masked position has P == 0 exactly; dP there was computed against cache rows nobody wrote - never touch it)
    float d = 0.f;
    for (int j = lane; j < Sk; j += 64) { const float pv = bf16_to_f32(p[j]); if (pv != 0.f) d += pv * dp[j]; }
    d = wave_sum(d);
    for (int j = lane; j < ldds; j += 64) {
        const float pv = j < Sk ? bf16_to_f32(p[j]) : 0.f;
        ds[j] = pv != 0.f ? f32_to_bf16(pv * (dp[j] - d) * scale) : (bf16_t)0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// RMSNorm backward (hf:models/mistral/modeling_mistral.py:182-199; the scale is frozen in Stage 1: dx only):
//   r = rsqrt(mean(x^2) + eps), xh = x r, g = dy w:  dx = r (g - xh mean(g xh)) [+ dres]
// One wave per row of `cols` <= 8192.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, long rows, int cols, float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* xr = x + row * cols;
    const bf16_t* gr = dy + row * cols;
    float ss = 0.f, gx = 0.f;
    for (int c = lane; c < cols; c += 64) { const float xv = bf16_to_f32(xr[c]); ss += xv * xv; gx += bf16_to_f32(gr[c]) * w[c] * xv; }
    ss = wave_sum(ss); gx = wave_sum(gx);
    const float r = rsqrtf(ss / cols + eps);
    const float k = gx * r * r / cols;            // mean(g xh) r = sum(g x) r^2 / cols ... applied to xh = x r below
    for (int c = lane; c < cols; c += 64) {
        const float xv = bf16_to_f32(xr[c]);
        float v = r * (bf16_to_f32(gr[c]) * w[c] - xv * k);
        if (dres) v += bf16_to_f32(dres[row * cols + c]);
        dx[row * cols + c] = f32_to_bf16(v);
    }
}

// cols = NCH * 512 (Zephyr: 4096): the row of x, dy and w in registers from one batch of 16-byte loads (the loop form above moves 2 bytes
// per lane per access, twice over the row).  The per-lane partial sums run over other elements than the loop form's, so the fp32 sums
// differ in the last bits from it - deterministically (same order every run).
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_rows_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ dy,
                                                               const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, long rows, float eps) {
    constexpr int cols = NCH * 512;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* xr = x + row * cols + lane * 8;
    const bf16_t* gr = dy + row * cols + lane * 8;
    u16x8 xv[NCH], gv[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { xv[c] = *(const u16x8*)(xr + c * 512); gv[c] = *(const u16x8*)(gr + c * 512); }
    float gw[NCH][8];
    float ss = 0.f, gx = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const f32x4 w0 = *(const f32x4*)(w + c * 512 + lane * 8), w1 = *(const f32x4*)(w + c * 512 + lane * 8 + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = bf16_to_f32(xv[c][j]);
            gw[c][j] = bf16_to_f32(gv[c][j]) * (j < 4 ? w0[j] : w1[j - 4]);
            ss += xf * xf; gx += gw[c][j] * xf;
        }
    }
    ss = wave_sum(ss); gx = wave_sum(gx);
    const float r = rsqrtf(ss / cols + eps);
    const float k = gx * r * r / cols;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        u16x8 dr = {0, 0, 0, 0, 0, 0, 0, 0};
        if (dres) dr = *(const u16x8*)(dres + row * cols + c * 512 + lane * 8);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = r * (gw[c][j] - bf16_to_f32(xv[c][j]) * k);
            if (dres) v += bf16_to_f32(dr[j]);
            o[j] = f32_to_bf16(v);
        }
        *(u16x8*)(dx + row * cols + c * 512 + lane * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// LayerNorm backward with parameter gradients (torch.nn.LayerNorm of the Q-Former, ref:...multimodal_projector/builder.py:14-27,68-70).
//   xh = (x - mean) rstd, g = dy w:   dx = rstd (g - mean(g) - xh mean(g xh)) [+ dres];   dw = sum_rows dy xh;   db = sum_rows dy
// Two kernels (round 2, second form: the first - one workgroup per row with four block reductions - took 1.7 ms per call, 13.5 % of the
// Stage-1 step, profiles/r02_train_kernel_stats.txt):
//   ln_bwd_dx_kernel    one WAVE per row (the 8-10 KiB row is re-read from L1 for each of its four sums: no LDS, no barrier); writes dx
//                       (optional) and the row's (mean, rstd) for the second kernel
//   ln_bwd_dwdb_kernel  thread = one column, workgroup (x, g) walks rows g, g + G, ... (512-byte coalesced row pieces) accumulating
//                       dw / db in registers -> part[g][2][cols], added in workgroup order by layernorm_bwd_reduce_kernel (no float
//                       atomics: gradients are reproducible bit for bit)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const bf16_t* __restrict__ dy,
                                                        const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ stats,
                                                        long rows, int cols, float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const bf16_t* xr = x + row * cols;
    const bf16_t* gr = dy + row * cols;
    float s = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        const u16x8 v = *(const u16x8*)(xr + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += bf16_to_f32(v[j]);
    }
    const float mean = wave_sum(s) / cols;
    float vs = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        const u16x8 v = *(const u16x8*)(xr + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = bf16_to_f32(v[j]) - mean; vs += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(vs) / cols + eps);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    if (!dx) return;
    float sg = 0.f, sgx = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
        const u16x8 xv = *(const u16x8*)(xr + c), gv = *(const u16x8*)(gr + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = bf16_to_f32(gv[j]) * w[c + j];
            sg += g; sgx += g * (bf16_to_f32(xv[j]) - mean) * rstd;
        }
    }
    const float mg = wave_sum(sg) / cols, mgx = wave_sum(sgx) / cols;
    for (int c = lane * 8; c < cols; c += 512) {
        const u16x8 xv = *(const u16x8*)(xr + c), gv = *(const u16x8*)(gr + c);
        u16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
        if (dres) rv = *(const u16x8*)(dres + row * cols + c);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = rstd * (bf16_to_f32(gv[j]) * w[c + j] - mg - (bf16_to_f32(xv[j]) - mean) * rstd * mgx);
            if (dres) v += bf16_to_f32(rv[j]);
            o[j] = f32_to_bf16(v);
        }
        *(u16x8*)(dx + row * cols + c) = o;
    }
}
__global__ __launch_bounds__(256) void ln_bwd_dwdb_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, const float* __restrict__ stats,
                                                          float* __restrict__ part, long rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float dw = 0.f, db = 0.f;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) {
        const float g = bf16_to_f32(dy[r * cols + c]);
        dw += g * (bf16_to_f32(x[r * cols + c]) - stats[2 * r]) * stats[2 * r + 1];
        db += g;
    }
    float* pw = part + (size_t)blockIdx.y * 2 * cols;
    pw[c] = dw; pw[cols + c] = db;
}
// dw[c] += sum_g part[g][0][c], db[c] += sum_g part[g][1][c]   (fixed order)
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ part, int G, int cols, float* __restrict__ dw,
                                                                   float* __restrict__ db) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= 2 * cols) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * 2 * cols + c];
    if (c < cols) dw[c] += s; else db[c - cols] += s;
}

// ---------------------------------------------------------------------------------------------------------------------------
// element-wise derivatives
// ---------------------------------------------------------------------------------------------------------------------------
// exact GELU forward on a saved pre-activation (training keeps h, the inference epilogue fuses it away) and its backward
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f32_to_bf16(act_gelu_erf(bf16_to_f32(h[i])));
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dh, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = bf16_to_f32(h[i]);
        const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
        dh[i] = f32_to_bf16(bf16_to_f32(dy[i]) * (cdf + x * pdf));
    }
}
// SwiGLU: act = silu(g) u with the gate / up pre-activations in the GEMM's interleaved column order [16 g | 16 u]:
// gu [rows, 2I], dact [rows, I] -> dgu [rows, 2I] in the same interleaved order (= the rows of the packed gate|up weight)
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dact, bf16_t* __restrict__ dgu,
                                                         long rows, int I) {
    const long n = rows * I;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / I;
        const int c = (int)(i - row * I), grp = c >> 4, in = c & 15;
        const size_t o = (size_t)row * 2 * I + grp * 32 + in;
        const float g = bf16_to_f32(gu[o]), u = bf16_to_f32(gu[o + 16]), d = bf16_to_f32(dact[i]);
        const float sg = 1.f / (1.f + __expf(-g));
        dgu[o] = f32_to_bf16(d * u * sg * (1.f + g * (1.f - sg)));
        dgu[o + 16] = f32_to_bf16(d * g * sg);
    }
}
// the SwiGLU forward on saved pre-activations (training path)
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ act, long rows, int I) {
    const long n = rows * I;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / I;
        const int c = (int)(i - row * I), grp = c >> 4, in = c & 15;
        const size_t o = (size_t)row * 2 * I + grp * 32 + in;
        act[i] = f32_to_bf16(act_silu(bf16_to_f32(gu[o])) * bf16_to_f32(gu[o + 16]));
    }
}

// RoPE backward + re-assembly of the fused-QKV gradient: dqkv[row][Hq*D | Hkv*D | Hkv*D] from dq [rows, Hq, D] (gradient of the
// ROTATED queries), dk / dv [B, Hkv, Sk_ld, D] fp32 (gradient of the rotated keys / of the values, per cache position).
// The rotation is orthogonal: d(pre) = d(post) cos - rot_half(d(post)) sin   (hf:models/mistral/modeling_mistral.py:51-81 transposed)
__global__ __launch_bounds__(256) void rope_bwd_assemble_kernel(const bf16_t* __restrict__ dq, const float* __restrict__ dk, const float* __restrict__ dv,
                                                                bf16_t* __restrict__ dqkv, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                                const int* __restrict__ pos, int B, int S, int Hq, int Hkv, int D, int Sk_ld) {
    const int QKV = (Hq + 2 * Hkv) * D, half = D >> 1;
    const long n = (long)B * S * (Hq + 2 * Hkv) * half;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int d = (int)(i % half);
        long t = i / half;
        const int head = (int)(t % (Hq + 2 * Hkv));
        const long row = t / (Hq + 2 * Hkv);
        const int b = (int)(row / S), s = (int)(row - (long)b * S);
        bf16_t* out = dqkv + row * QKV + (size_t)head * D;
        if (head >= Hq + Hkv) {                 // V: no rotation
            const float* src = dv + (((size_t)b * Hkv + (head - Hq - Hkv)) * Sk_ld + s) * D;
            out[d] = f32_to_bf16(src[d]); out[d + half] = f32_to_bf16(src[d + half]);
            continue;
        }
        float a, bb;
        if (head < Hq) { const bf16_t* src = dq + (row * Hq + head) * D; a = bf16_to_f32(src[d]); bb = bf16_to_f32(src[d + half]); }
        else { const float* src = dk + (((size_t)b * Hkv + (head - Hq)) * Sk_ld + s) * D; a = src[d]; bb = src[d + half]; }
        const int p = pos[row];
        const float c = cosT[(size_t)p * half + d], sn = sinT[(size_t)p * half + d];
        // forward: y1 = x1 c - x2 s, y2 = x2 c + x1 s  ->  dx1 = dy1 c + dy2 s, dx2 = dy2 c - dy1 s
        out[d] = f32_to_bf16(a * c + bb * sn);
        out[d + half] = f32_to_bf16(bb * c - a * sn);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// causal-LM cross-entropy (hf:loss/loss_utils.py ForCausalLMLoss: fp32 logits, shift by one, ignore_index -100, mean over the
// valid targets): row r = (b, s) is scored against label[b][s + 1]; loss_rows[r] = lse - logit[target] (0 when ignored);
// dlogits[r][:] = (softmax - onehot) * inv_n (bf16, zero row when ignored, zero padding up to ldd).  One workgroup per row.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int V, const int* __restrict__ labels, int S,
                                                            float inv_n, float* __restrict__ loss_rows, bf16_t* __restrict__ dlogits, int ldd) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const int s = (int)(row % S), t = threadIdx.x;
    const int target = s + 1 < S ? labels[row + 1] : -100;
    bf16_t* d = dlogits ? dlogits + row * ldd : nullptr;         // null: loss only (the forward path of the drop-in model)
    if (target < 0 || target >= V) {
        if (d) for (int j = t; j < ldd; j += 256) d[j] = 0;
        if (t == 0) loss_rows[row] = 0.f;
        return;
    }
    const float* l = logits + row * V;
    float m = -INFINITY;
    for (int j = t; j < V; j += 256) m = fmaxf(m, l[j]);
    m = wave_max(m);
    if ((t & 63) == 0) red[t >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = t; j < V; j += 256) sum += __expf(l[j] - m);
    sum = block_sum_256(sum, red);
    const float inv = inv_n / sum;
    if (d) for (int j = t; j < ldd; j += 256) d[j] = j < V ? f32_to_bf16((__expf(l[j] - m) * inv) - (j == target ? inv_n : 0.f)) : (bf16_t)0;
    if (t == 0) loss_rows[row] = (m + logf(sum)) - l[target];
}

// mean of loss_rows over the rows whose target is valid (label[b][s + 1] in [0, V)): one workgroup, thread t adds rows t, t + 256, ...
// in order, then the 256 partials meet in a fixed order - out[0] = mean (NaN when nothing is valid, like torch), out[1] = valid count
__global__ __launch_bounds__(256) void loss_mean_kernel(const float* __restrict__ loss_rows, const int* __restrict__ labels, long rows, int S, int V,
                                                        float* __restrict__ out) {
    __shared__ float red[4];
    float sum = 0.f, cnt = 0.f;
    for (long r = threadIdx.x; r < rows; r += 256) {
        const int s = (int)(r % S);
        const int target = s + 1 < S ? labels[r + 1] : -100;
        if (target >= 0 && target < V) { sum += loss_rows[r]; cnt += 1.f; }
    }
    sum = block_sum_256(sum, red);
    __syncthreads();
    cnt = block_sum_256(cnt, red);
    if (threadIdx.x == 0) { out[0] = sum / cnt; out[1] = cnt; }       // no valid target: 0 / 0 = NaN, as torch's mean over nothing
}

// out[c] += sum_r y[r][c]   (bias gradients).  Stage 1: workgroup (x, g) adds rows g, g + G, ... of its 256 columns (512-byte row
// pieces) into part[g][c]; stage 2 adds the G partials in order.
__global__ __launch_bounds__(256) void colsum_part_kernel(const bf16_t* __restrict__ y, int ld, long rows, int cols, float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (long r = blockIdx.y; r < rows; r += gridDim.y) s += bf16_to_f32(y[r * ld + c]);
    part[(size_t)blockIdx.y * cols + c] = s;
}
// the same sums with 16-byte accesses (cols % 8 == 0, ld % 8 == 0, y 16-byte aligned): 8 columns per thread, four rows in flight
__global__ __launch_bounds__(256) void colsum_part8_kernel(const bf16_t* __restrict__ y, int ld, long rows, int cols, float* __restrict__ part) {
    const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= cols) return;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long G = gridDim.y;
    long r = blockIdx.y;
    for (; r + 3 * G < rows; r += 4 * G) {
        u16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const u16x8*)(y + (r + u * G) * ld + c);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += bf16_to_f32(v[u][j]);
    }
    for (; r < rows; r += G) {
        const u16x8 v = *(const u16x8*)(y + r * ld + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += bf16_to_f32(v[j]);
    }
    float* o = part + (size_t)blockIdx.y * cols + c;
    *(f32x4*)o = (f32x4){s[0], s[1], s[2], s[3]};
    *(f32x4*)(o + 4) = (f32x4){s[4], s[5], s[6], s[7]};
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, int G, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * cols + c];
    out[c] += s;
}

// rows of `src` picked by idx (>= 0) -> dst; dst[r] = 0 where idx[r] < 0   (visual-token rows out of d(inputs_embeds))
__global__ __launch_bounds__(256) void gather_rows_idx_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst,
                                                              long rows, int cols) {
    const int per_row = cols >> 3;
    const long n = rows * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / per_row;
        const int c = (int)(i - r * per_row) * 8;
        const int j = idx[r];
        u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j >= 0) v = *(const u16x8*)(src + (size_t)j * cols + c);
        *(u16x8*)(dst + r * cols + c) = v;
    }
}

// dst[r] (+)= sum of src rows: dst [n_dst, cols] bf16 <- for every r: sum over k < n_src with map[k] == r of src[k]   (block 0 of the
// Q-Former runs once per SAMPLE: the gradient of a sample's 32 rows is the sum over its tiles).  Small: one thread per element.
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ map, int n_src, int rows_per,
                                                               bf16_t* __restrict__ dst, int n_dst, int cols) {
    const long n = (long)n_dst * rows_per * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cols);
        const long t = i / cols;
        const int q = (int)(t % rows_per), r = (int)(t / rows_per);
        float s = 0.f;
        for (int k = 0; k < n_src; ++k) if (map[k] == r) s += bf16_to_f32(src[((size_t)k * rows_per + q) * cols + c]);
        dst[i] = f32_to_bf16(s);
    }
}

__global__ __launch_bounds__(256) void axpy_f32_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += x[i];
}
__global__ __launch_bounds__(256) void add_bf16_kernel(bf16_t* __restrict__ y, const bf16_t* __restrict__ x, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f32_to_bf16(bf16_to_f32(y[i]) + bf16_to_f32(x[i]));
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = bf16_to_f32(x[i]);
}
// out[q][c] += sum_b src[b * stride + q * cols + c]   (fp32 accumulation of a few bf16 row blocks: the learned_queries gradient)
__global__ __launch_bounds__(256) void acc_rows_f32_kernel(float* __restrict__ out, const bf16_t* __restrict__ src, int n_batches, long stride, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float s = 0.f;
        for (int b = 0; b < n_batches; ++b) s += bf16_to_f32(src[(size_t)b * stride + i]);
        out[i] += s;
    }
}
__global__ __launch_bounds__(256) void f32_rows_to_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = f32_to_bf16(x[i]);
}

// torch.optim.AdamW (no amsgrad), one tensor: fp32 master / moments / gradient; writes the engine's working copy (bf16 matrix or
// fp32 vector) and clears the gradient for the next accumulation.  hf Trainer: betas 0.9 / 0.999, eps 1e-8 (ref:script/pretrain.sh:39-42)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float* __restrict__ g,
                                                    void* __restrict__ work, int work_bf16, long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2_sqrt) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi; m[i] = mi; v[i] = vi; g[i] = 0.f;
        if (work_bf16) ((bf16_t*)work)[i] = f32_to_bf16(pi); else ((float*)work)[i] = pi;
    }
}

inline int grid_for(long n, int cap = 8192) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > cap ? cap : b)); }

}  // namespace

int vz_launch_transpose(const bf16_t* src, long src_rs, long src_so, long src_si, bf16_t* dst, long dst_rs, long dst_so, long dst_si, int R,
                        int C, int n_outer, int n_inner, int col0, hipStream_t s) {
    VZ_CHECK_ARG(src && dst && R > 0 && C > 0 && n_outer > 0 && n_inner > 0 && (long)n_outer * n_inner <= 65535, "transpose: bad argument");
    const long strides[] = {src_rs, src_so, src_si, dst_rs, dst_so, dst_si, (long)R, (long)C, (long)col0};
    bool wide = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    for (long v : strides) wide = wide && (v & 7) == 0;
    if (wide && R >= 1024 && C >= 1024)
        hipLaunchKernelGGL((transpose64_kernel<128>), dim3((C + 127) / 128, (R + 127) / 128, n_outer * n_inner), dim3(256), 0, s, src, src_rs, src_so, src_si,
                           dst, dst_rs, dst_so, dst_si, R, C, n_inner, col0);
    else if (wide)
        hipLaunchKernelGGL((transpose64_kernel<64>), dim3((C + 63) / 64, (R + 63) / 64, n_outer * n_inner), dim3(256), 0, s, src, src_rs, src_so, src_si, dst,
                           dst_rs, dst_so, dst_si, R, C, n_inner, col0);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, n_outer * n_inner), dim3(256), 0, s, src, src_rs, src_so, src_si, dst,
                           dst_rs, dst_so, dst_si, R, C, n_inner, col0);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_softmax_fwd(const float* S, int lds_, bf16_t* P, int ldp, long rows, int H, int Sq, int Sk, float scale, int causal, int window,
                          const int* kv_len, hipStream_t s) {
    VZ_CHECK_ARG(S && P && rows > 0 && Sk <= lds_ && Sk <= ldp, "softmax_fwd: bad argument");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, S, lds_, P, ldp, rows, H, Sq, Sk, scale, causal, window, kv_len);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_softmax_bwd(const bf16_t* P, int ldp, const float* dP, int lddp, bf16_t* dS, int ldds, long rows, int Sk, float scale, hipStream_t s) {
    VZ_CHECK_ARG(P && dP && dS && rows > 0, "softmax_bwd: bad argument");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, P, ldp, dP, lddp, dS, ldds, rows, Sk, scale);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_rmsnorm_bwd(const bf16_t* x, const float* w, const bf16_t* dy, const bf16_t* dres, bf16_t* dx, long rows, int cols, float eps,
                          hipStream_t s) {
    VZ_CHECK_ARG(x && w && dy && dx && rows > 0 && cols > 0, "rmsnorm_bwd: bad argument");
    const bool aligned = (((uintptr_t)x | (uintptr_t)w | (uintptr_t)dy | (uintptr_t)dres | (uintptr_t)dx) & 15) == 0;
    if (cols == 4096 && aligned)
        hipLaunchKernelGGL((rmsnorm_bwd_rows_kernel<8>), dim3((int)((rows + 3) / 4)), dim3(256), 0, s, x, w, dy, dres, dx, rows, eps);
    else
        hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, x, w, dy, dres, dx, rows, cols, eps);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
// `part` holds G * 2 * cols floats (G = vz_layernorm_bwd_groups(rows)) followed by 2 * rows floats of row statistics
int vz_layernorm_bwd_groups(long rows) { return (int)(rows < 128 ? rows : 128); }
size_t vz_layernorm_bwd_scratch_floats(long rows, int cols) { return (size_t)vz_layernorm_bwd_groups(rows) * 2 * cols + 2 * (size_t)rows; }
int vz_launch_layernorm_bwd(const bf16_t* x, const float* w, const bf16_t* dy, const bf16_t* dres, bf16_t* dx, float* part, float* dw, float* db,
                            long rows, int cols, float eps, hipStream_t s) {
    VZ_CHECK_ARG(x && w && dy && part && dw && db && rows > 0 && cols > 0 && (cols & 7) == 0, "layernorm_bwd: bad argument (cols %% 8)");
    const int G = vz_layernorm_bwd_groups(rows);
    float* stats = part + (size_t)G * 2 * cols;
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, x, w, dy, dres, dx, stats, rows, cols, eps);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_dwdb_kernel, dim3((cols + 255) / 256, G), dim3(256), 0, s, x, dy, stats, part, rows, cols);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((2 * cols + 255) / 256), dim3(256), 0, s, part, G, cols, dw, db);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_gelu_fwd(const bf16_t* h, bf16_t* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, h, y, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_gelu_bwd(const bf16_t* h, const bf16_t* dy, bf16_t* dh, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, h, dy, dh, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_swiglu_fwd(const bf16_t* gu, bf16_t* act, long rows, int I, hipStream_t s) {
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * I)), dim3(256), 0, s, gu, act, rows, I);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_swiglu_bwd(const bf16_t* gu, const bf16_t* dact, bf16_t* dgu, long rows, int I, hipStream_t s) {
    VZ_CHECK_ARG(I % 16 == 0, "swiglu_bwd: I %% 16");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * I)), dim3(256), 0, s, gu, dact, dgu, rows, I);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_rope_bwd_assemble(const bf16_t* dq, const float* dk, const float* dv, bf16_t* dqkv, const float* cosT, const float* sinT, const int* pos,
                                int B, int S, int Hq, int Hkv, int D, int Sk_ld, hipStream_t s) {
    hipLaunchKernelGGL(rope_bwd_assemble_kernel, dim3(grid_for((long)B * S * (Hq + 2 * Hkv) * (D / 2))), dim3(256), 0, s, dq, dk, dv, dqkv, cosT, sinT,
                       pos, B, S, Hq, Hkv, D, Sk_ld);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_cross_entropy(const float* logits, int V, const int* labels, long rows, int S, float inv_n, float* loss_rows, bf16_t* dlogits, int ldd,
                            hipStream_t s) {
    VZ_CHECK_ARG(logits && labels && loss_rows && rows > 0 && (!dlogits || ldd >= V), "cross_entropy: bad argument");
    hipLaunchKernelGGL(cross_entropy_kernel, dim3((int)rows), dim3(256), 0, s, logits, V, labels, S, inv_n, loss_rows, dlogits, ldd);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
// hf:loss/loss_utils.py ForCausalLMLoss on fp32 logits [B, S, V] and labels [B, S] (ignore_index -100, shift by one, mean over
// the valid targets): d_loss_rows = B * S floats of scratch, d_out[0] = loss, d_out[1] = number of valid targets
int vz_launch_causal_lm_loss(const float* logits, int B, int S, int V, const int* labels, float* loss_rows, float* out, hipStream_t s) {
    VZ_CHECK_ARG(logits && labels && loss_rows && out && B > 0 && S > 0 && V > 0, "causal_lm_loss: bad argument");
    const long rows = (long)B * S;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3((int)rows), dim3(256), 0, s, logits, V, labels, S, 1.0f, loss_rows, (bf16_t*)nullptr, 0);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(256), 0, s, (const float*)loss_rows, labels, rows, S, V, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
// `part`: vz_colsum_groups(rows) * cols floats of scratch
int vz_colsum_groups(long rows) { return (int)(rows < 4096 ? (rows + 63) / 64 : 256); }
int vz_launch_colsum(const bf16_t* y, int ld, long rows, int cols, float* part, float* out, hipStream_t s) {
    VZ_CHECK_ARG(y && part && out && rows > 0 && cols > 0, "colsum: bad argument");
    const int G = vz_colsum_groups(rows);
    if ((cols & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)part & 15) == 0)
        hipLaunchKernelGGL(colsum_part8_kernel, dim3((cols / 8 + 255) / 256, G), dim3(256), 0, s, y, ld, rows, cols, part);
    else
        hipLaunchKernelGGL(colsum_part_kernel, dim3((cols + 255) / 256, G), dim3(256), 0, s, y, ld, rows, cols, part);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((cols + 255) / 256), dim3(256), 0, s, part, G, cols, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_gather_rows_idx(const bf16_t* src, const int* idx, bf16_t* dst, long rows, int cols, hipStream_t s) {
    VZ_CHECK_ARG(cols % 8 == 0, "gather_rows_idx: cols %% 8");
    hipLaunchKernelGGL(gather_rows_idx_kernel, dim3(grid_for(rows * (cols / 8))), dim3(256), 0, s, src, idx, dst, rows, cols);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_segment_sum_rows(const bf16_t* src, const int* map, int n_src, int rows_per, bf16_t* dst, int n_dst, int cols, hipStream_t s) {
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(grid_for((long)n_dst * rows_per * cols)), dim3(256), 0, s, src, map, n_src, rows_per, dst, n_dst, cols);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_axpy_f32(float* y, const float* x, long n, hipStream_t s) {
    hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_add_bf16(bf16_t* y, const bf16_t* x, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, y, x, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_bf16_to_f32(const bf16_t* x, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_acc_rows_f32(float* out, const bf16_t* src, int n_batches, long stride, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(acc_rows_f32_kernel, dim3(grid_for((long)rows * cols)), dim3(256), 0, s, out, src, n_batches, stride, (long)rows * cols);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_f32_to_bf16(const float* x, bf16_t* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(f32_rows_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
int vz_launch_adamw(float* p, float* m, float* v, float* g, void* work, int work_bf16, long n, float lr, float b1, float b2, float eps, float wd,
                    int t, hipStream_t s) {
    VZ_CHECK_ARG(p && m && v && g && work && n > 0 && t >= 1, "adamw: bad argument");
    const float bc1 = 1.f - powf(b1, (float)t), bc2s = sqrtf(1.f - powf(b2, (float)t));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, m, v, g, work, work_bf16, n, lr, b1, b2, eps, wd, bc1, bc2s);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
