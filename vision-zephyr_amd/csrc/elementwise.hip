// HBM-bound kernels of the Vision-Zephyr hot path (gfx950): norms, RoPE + KV-cache append,
// embedding gather / [vision;text] splice, CLIP patch im2col + token assembly, multi-layer
// fusion, argmax.  All of them move 16 bytes per lane per access (8 bf16), one wave per row
// where rows are 2-10 KiB, and do their arithmetic in fp32 with a single rounding to bf16.
#include <algorithm>

#include "vz_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm: one wave per row, row cached in registers (cols <= 5120 = 10 chunks of 512)
//   torch.nn.LayerNorm (biased variance, eps inside the sqrt): hf:models/clip/modeling_clip.py:353-384,
//   ref:vis_zephyr/model/multimodal_projector/builder.py:14-27,68-70
//   MistralRMSNorm: hf:models/mistral/modeling_mistral.py:182-199
// ------------------------------------------------------------------------------------------------
constexpr int NORM_MAX_CHUNKS = 10;

template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                   const float* __restrict__ w, const float* __restrict__ b, int rows,
                                                   int cols, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    const int nch = (cols + 511) >> 9;
    float v[NORM_MAX_CHUNKS][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int k = c * 512 + lane * 8;
        if (c < nch && k < cols) {
            const u16x8 t = *(const u16x8*)(xr + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = bf16_to_f32(t[j]); s += RMS ? v[c][j] * v[c][j] : v[c][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
            const int k = c * 512 + lane * 8;
            if (c < nch && k < cols) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; q += d * d; }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int c = 0; c < NORM_MAX_CHUNKS; ++c) {
        const int k = c * 512 + lane * 8;
        if (c < nch && k < cols) {
            const f32x4 w0 = *(const f32x4*)(w + k), w1 = *(const f32x4*)(w + k + 4);
            f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
            if (!RMS) { b0 = *(const f32x4*)(b + k); b1 = *(const f32x4*)(b + k + 4); }
            u16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wj = j < 4 ? w0[j] : w1[j - 4];
                const float bj = j < 4 ? b0[j] : b1[j - 4];
                const float r = RMS ? wj * (v[c][j] * rstd) : (v[c][j] - mean) * rstd * wj + bj;
                o[j] = f32_to_bf16(r);
            }
            *(u16x8*)(yr + k) = o;
        }
    }
}

// The widths on the hot path (1024 CLIP, 4096 Zephyr / Q-Former, 5120 fused features) are whole 512-element chunks: without the
// per-chunk bounds test every load of a row is issued before the first is used.  (The generic kernel above compiles to one
// `global_load ; s_waitcnt vmcnt(0)` PER CHUNK inside its own exec branch - eight dependent round trips for a Zephyr row: 10.3 us for
// 2048 x 4096, found in the ISA in the third session of round 2.)  Same arithmetic in the same order: bit-identical results.
template <bool RMS, int NCH>
__global__ __launch_bounds__(256) void norm_rows_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                                        const float* __restrict__ w, const float* __restrict__ b, int rows, float eps) {
    constexpr int cols = NCH * 512;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (size_t)row * ldx + lane * 8;
    u16x8 t[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) t[c] = *(const u16x8*)(xr + c * 512);
    // the scale vector does not wait for the row statistics: requested here, behind the row, it arrives under the reduction
    const float* wr = w + lane * 8;
    f32x4 w0[NCH], w1[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { w0[c] = *(const f32x4*)(wr + c * 512); w1[c] = *(const f32x4*)(wr + c * 512 + 4); }
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[c][j] = bf16_to_f32(t[c][j]); s += RMS ? v[c][j] * v[c][j] : v[c][j]; }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
    } else {
        mean = s / (float)cols;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[c][j] - mean; q += d * d; }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)cols + eps);
    }
    bf16_t* yr = y + (size_t)row * ldy + lane * 8;
    const float* br = RMS ? nullptr : b + lane * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        if (!RMS) { b0 = *(const f32x4*)(br + c * 512); b1 = *(const f32x4*)(br + c * 512 + 4); }
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wj = j < 4 ? w0[c][j] : w1[c][j - 4];
            const float bj = j < 4 ? b0[j] : b1[j - 4];
            const float r = RMS ? wj * (v[c][j] * rstd) : (v[c][j] - mean) * rstd * wj + bj;
            o[j] = f32_to_bf16(r);
        }
        *(u16x8*)(yr + c * 512) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// RoPE (rotate-half, hf:models/mistral/modeling_mistral.py:51-81) on Q and K of a fused QKV row,
// K/V appended to the cache [B][Hkv][max_ctx][D].  D = 128; 16 lanes x 16 bytes per (token, head).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kv_kernel(const bf16_t* __restrict__ qkv, int ld, bf16_t* __restrict__ q_out,
                                                      bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                      const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                      const int* __restrict__ pos, const int* __restrict__ slot, int B,
                                                      int S, int Hq, int Hkv, int D, int max_ctx) {
    // 16 lanes per (token, head): lane `sub` owns the 8 elements d0 = (sub>>3)*64 + (sub&7)*8 .. +7 (one 16-byte access);
    // the rotation partner (d +- 64) lives in lane sub ^ 8.  Four items per wave, 16 per workgroup.
    const int lane = threadIdx.x & 63, sub = lane & 15;
    // q_out == nullptr: the queries are rotated by their consumer (the prefill attention's Q load) - only the K and V heads are items
    const int heads = q_out ? Hq + 2 * Hkv : 2 * Hkv, h_first = q_out ? 0 : Hq;
    const long item = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const long n_items = (long)B * S * heads;
    const bool live = item < n_items;
    const long it = live ? item : n_items - 1;          // keep every lane in the shuffles below
    const int tok = (int)(it / heads), h = h_first + (int)(it % heads);
    const int b = tok / S;
    const int d0 = (sub >> 3) * 64 + (sub & 7) * 8;
    const bf16_t* src = qkv + (size_t)tok * ld + (size_t)h * D + d0;
    const uint4 raw = *(const uint4*)src;
    const int sl = slot[tok];
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    unsigned pw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pw[i] = __shfl_xor(w[i], 8, 64);
    if (!live) return;
    if (h >= Hq + Hkv) {  // V: plain copy into the cache
        if (sl >= 0) *(uint4*)(vc + (((size_t)b * Hkv + (h - Hq - Hkv)) * max_ctx + sl) * D + d0) = raw;
        return;
    }
    bf16_t* dst;
    if (h < Hq) dst = q_out + ((size_t)tok * Hq + h) * D + d0;
    else {
        if (sl < 0) return;
        dst = kc + (((size_t)b * Hkv + (h - Hq)) * max_ctx + sl) * D + d0;
    }
    const int p = pos[tok];
    const float* cp = cosT + (size_t)p * (D / 2) + (sub & 7) * 8;
    const float* sp = sinT + (size_t)p * (D / 2) + (sub & 7) * 8;
    const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4), s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
    const float sgn = (sub >> 3) ? 1.f : -1.f;           // out[d] = x*cos - y*sin ; out[d+64] = y'*cos + x'*sin
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = bf16_to_f32(w[i] & 0xFFFF), x1 = bf16_to_f32(w[i] >> 16);
        const float y0 = bf16_to_f32(pw[i] & 0xFFFF), y1 = bf16_to_f32(pw[i] >> 16);
        const float ca = i < 2 ? c0[2 * i] : c1[2 * i - 4], cb = i < 2 ? c0[2 * i + 1] : c1[2 * i - 3];
        const float sa = i < 2 ? s0[2 * i] : s1[2 * i - 4], sb = i < 2 ? s0[2 * i + 1] : s1[2 * i - 3];
        o[i] = pack_bf16x2(x0 * ca + sgn * (y0 * sa), x1 * cb + sgn * (y1 * sb));
    }
    *(uint4*)dst = make_uint4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------------------------------------
// Row gather: embedding lookup and the [vision ; text] splice (a6/a7 data movement,
// ref:vis_zephyr/model/vis_zephyr_arch.py:236-305,476-530).  One wave per 16-byte-chunked row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const int* __restrict__ kind, const int* __restrict__ idx, int rows,
                                                          int cols, const bf16_t* __restrict__ table,
                                                          const bf16_t* __restrict__ visual, bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int kd = kind ? kind[row] : 0;
    const bf16_t* src = kd == 0 ? table + (size_t)idx[row] * cols : kd == 1 ? visual + (size_t)idx[row] * cols : nullptr;
    bf16_t* dst = out + (size_t)row * cols;
    // four 16-byte pieces requested before the first is stored (a load-store loop waits for every piece in turn)
    for (int k0 = lane * 8; k0 < cols; k0 += 2048) {
        uint4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = (src && k0 + u * 512 < cols) ? *(const uint4*)(src + k0 + u * 512) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * 512 < cols) *(uint4*)(dst + k0 + u * 512) = t[u];
    }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, long src_stride, bf16_t* __restrict__ dst,
                                                        long dst_stride, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* sr = src + (size_t)row * src_stride;
    bf16_t* dr = dst + (size_t)row * dst_stride;
    for (int k0 = lane * 8; k0 < cols; k0 += 2048) {
        uint4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = k0 + u * 512 < cols ? *(const uint4*)(sr + k0 + u * 512) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * 512 < cols) *(uint4*)(dr + k0 + u * 512) = t[u];
    }
}

// The first len[z] tokens of KV-cache row src[z] -> row dst[z], every layer, K and V, every KV head in one launch
// (cache: [layer][K|V][row][kv head][max_ctx][D]).  grid (4-KiB pieces of a head's len * D * 2 bytes, layers * 2 * Hkv, moves).
__global__ __launch_bounds__(256) void kv_move_rows_kernel(bf16_t* __restrict__ kv, size_t layer_elems, int max_batch, int Hkv,
                                                           int max_ctx, int D, KvMoves mv) {
    const int z = blockIdx.z, y = blockIdx.y;
    const int head = y % Hkv, half = (y / Hkv) & 1, layer = y / (2 * Hkv);
    const size_t row_elems = (size_t)Hkv * max_ctx * D;
    bf16_t* base = kv + (size_t)layer * layer_elems + (size_t)half * (layer_elems / 2) + (size_t)head * max_ctx * D;
    const bf16_t* src = base + (size_t)mv.src[z] * row_elems;
    bf16_t* dst = base + (size_t)mv.dst[z] * row_elems;
    const long n = (long)mv.len[z] * D;
    const long k = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (k < n) *(uint4*)(dst + k) = *(const uint4*)(src + k);
}

// ------------------------------------------------------------------------------------------------
// CLIP patch embedding front end (hf:models/clip/modeling_clip.py:148-154,202-219):
// im2col of the 14x14x3 patches into [T*576, kpad] (k = c*196 + py*14 + px, zero padded to kpad),
// so that the Conv2d(3->1024, k14, s14, no bias) becomes the tile GEMM; then class token + positions.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* __restrict__ img, int T, int image, int patch, int kpad,
                                                     bf16_t* __restrict__ out) {
    const int g = image / patch;
    const long total = (long)T * g * g * kpad;
    const int kreal = 3 * patch * patch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % kpad);
        const long rowi = i / kpad;
        bf16_t v = 0;
        if (k < kreal) {
            const int c = k / (patch * patch), rem = k % (patch * patch), py = rem / patch, px = rem % patch;
            const int gx = (int)(rowi % g), gy = (int)((rowi / g) % g), t = (int)(rowi / ((long)g * g));
            v = img[(((size_t)t * 3 + c) * image + gy * patch + py) * image + gx * patch + px];
        }
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void clip_assemble_kernel(const bf16_t* __restrict__ patch_out, const bf16_t* __restrict__ cls,
                                                            const bf16_t* __restrict__ pos, int T, int tokens, int C,
                                                            bf16_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T * tokens) return;
    const int t = row / tokens, tk = row % tokens;
    const bf16_t* src = tk == 0 ? cls : patch_out + ((size_t)t * (tokens - 1) + tk - 1) * C;
    const bf16_t* pp = pos + (size_t)tk * C;
    for (int k = lane * 8; k < C; k += 512) {
        const u16x8 a = *(const u16x8*)(src + k), p8 = *(const u16x8*)(pp + k);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16(bf16_to_f32(a[j]) + bf16_to_f32(p8[j]));
        *(u16x8*)(out + (size_t)row * C + k) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-layer fusion (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:58-78,
// ref:vis_zephyr/model/gating_fusion/gating_fusion.py:22-50): hidden states first_layer..last,
// drop (`skip` = 1, 'patch') or keep (`skip` = 0, 'cls_patch') CLS, `groups` means of `per_group` consecutive layers + the last
// layer, channel-concat.
// hs_base: [(layers), T, tokens, C] with `layer_stride` elements between layers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fusion_kernel(const bf16_t* __restrict__ hs_base, long layer_stride, int first_layer,
                                                     int groups, int per_group, int T, int tokens, int C, int skip,
                                                     bf16_t* __restrict__ out) {
    const int chunks_per_row = C / 8;
    const int keep = tokens - skip;
    const long total = (long)T * keep * (groups + 1) * chunks_per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % chunks_per_row);
        long r = i / chunks_per_row;
        const int gidx = (int)(r % (groups + 1));
        r /= (groups + 1);
        const int ptk = (int)(r % keep), t = (int)(r / keep);
        const size_t off = ((size_t)t * tokens + skip + ptk) * C + ch * 8;
        u16x8 o;
        if (gidx == groups) {
            o = *(const u16x8*)(hs_base + (size_t)(first_layer + groups * per_group) * layer_stride + off);
        } else {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int l = 0; l < per_group; ++l) {
                const u16x8 a = *(const u16x8*)(hs_base + (size_t)(first_layer + gidx * per_group + l) * layer_stride + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32(a[j]);
            }
            const float cnt = (float)per_group;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16(acc[j] / cnt);
        }
        *(u16x8*)(out + ((size_t)t * keep + ptk) * ((size_t)(groups + 1) * C) + (size_t)gidx * C + ch * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// argmax over fp32 logits (first maximal index, as torch.argmax), one block per row.  In the decode
// loop the same kernel is the step's tail: it publishes the token for the next step, appends it to
// the output ids and advances the per-slot position / context length kept on the device.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, int cols, int* __restrict__ ids,
                                                      int* __restrict__ pos, int* __restrict__ slot, int* __restrict__ len,
                                                      int* __restrict__ out_ids, int out_stride, const int* __restrict__ step,
                                                      int max_ctx, int rope_max, int* __restrict__ ring, int ring_n) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* lr = logits + (size_t)row * cols;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if ((cols & 3) == 0 && (((uintptr_t)lr) & 15) == 0) {
        // 16 bytes per lane, eight requests in flight per thread (a 32000-wide row is one batch): the scalar loop below waits for
        // every 4-byte load before it issues the next.  First maximum wins whatever the order of the comparisons.
        const int n4 = cols >> 2;
        for (int q0 = tid; q0 < n4; q0 += 8 * 1024) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = q0 + u * 1024 < n4 ? *(const f32x4*)(lr + (size_t)(q0 + u * 1024) * 4) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = t[u][j];
                    const int k = (q0 + u * 1024) * 4 + j;
                    if (v > bv || (v == bv && k < bi)) { bv = v; bi = k; }
                }
        }
    } else {
        for (int k = tid; k < cols; k += 1024) {
            const float v = lr[k];
            if (v > bv || (v == bv && k < bi)) { bv = v; bi = k; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;  // all-NaN row
        ids[row] = bi;
        if (out_ids) out_ids[(size_t)row * out_stride + (step ? *step : 0)] = bi;
        if (ring) ring[(size_t)row * ring_n + ((unsigned)step[1] % (unsigned)ring_n)] = bi;     // host-visible ring, slot = draw counter
        // saturating advance: a live row is kept inside the cache by the host's capacity check (vz_llm_decode_steps); a parked row
        // of a continuous batch steps for ever and must stay inside its own cache row / the rotary tables
        if (pos && pos[row] + 1 < rope_max) pos[row] += 1;
        if (len && len[row] < max_ctx) { len[row] += 1; if (slot) slot[row] += 1; }
    }
}

// step[0] = index of the step inside this vz_llm_decode_steps call; step[1] = tokens drawn since vz_llm_decode_begin (the
// Philox counter of the sampling tail)
__global__ void step_advance_kernel(int* step) { step[0] += 1; step[1] += 1; }

// vocab-parallel logits after the all-gather: gathered[r][row][j] (j < Vp, zero-padded shards) -> out[row][r*Vp + j]
__global__ __launch_bounds__(256) void repack_logits_kernel(const float* __restrict__ g, float* __restrict__ out, int rows, int Vp, int V, int tp) {
    const long total = (long)rows * V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int row = (int)(i / V), v = (int)(i % V);
        const int r = v / Vp, j = v - r * Vp;
        out[i] = g[((size_t)r * rows + row) * Vp + j];
    }
}

}  // namespace

static inline int rows_grid(long rows) { return (int)((rows + 3) / 4); }

int vz_launch_layernorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const float* w, const float* b, int rows, int cols,
                        float eps, hipStream_t s) {
    VZ_CHECK_ARG(x && y && w && b && rows > 0, "layernorm: bad argument");
    VZ_CHECK_ARG(cols % 8 == 0 && cols <= NORM_MAX_CHUNKS * 512 && ldx % 8 == 0 && ldy % 8 == 0,
                 "layernorm: cols=%d must be a multiple of 8 and <= %d", cols, NORM_MAX_CHUNKS * 512);
    if (cols == 1024) hipLaunchKernelGGL((norm_rows_kernel<false, 2>), dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, eps);
    else if (cols == 4096) hipLaunchKernelGGL((norm_rows_kernel<false, 8>), dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, eps);
    else if (cols == 5120) hipLaunchKernelGGL((norm_rows_kernel<false, 10>), dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, eps);
    else hipLaunchKernelGGL(norm_kernel<false>, dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, b, rows, cols, eps);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_rmsnorm(const bf16_t* x, int ldx, bf16_t* y, int ldy, const float* w, int rows, int cols, float eps,
                      hipStream_t s) {
    VZ_CHECK_ARG(x && y && w && rows > 0, "rmsnorm: bad argument");
    VZ_CHECK_ARG(cols % 8 == 0 && cols <= NORM_MAX_CHUNKS * 512 && ldx % 8 == 0 && ldy % 8 == 0,
                 "rmsnorm: cols=%d must be a multiple of 8 and <= %d", cols, NORM_MAX_CHUNKS * 512);
    if (cols == 4096) hipLaunchKernelGGL((norm_rows_kernel<true, 8>), dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, (const float*)nullptr, rows, eps);
    else hipLaunchKernelGGL(norm_kernel<true>, dim3(rows_grid(rows)), dim3(256), 0, s, x, ldx, y, ldy, w, (const float*)nullptr,
                            rows, cols, eps);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_rope_kv(const bf16_t* qkv, int ld, bf16_t* q_out, bf16_t* kc, bf16_t* vc, const float* cosT, const float* sinT,
                      const int* pos, const int* slot, int B, int S, int Hq, int Hkv, int D, int max_ctx, hipStream_t s) {
    VZ_CHECK_ARG(D == 128, "rope: head_dim %d unsupported (Zephyr uses 128)", D);
    const long items = (long)B * S * (q_out ? Hq + 2 * Hkv : 2 * Hkv);
    hipLaunchKernelGGL(rope_kv_kernel, dim3((int)((items + 15) / 16)), dim3(256), 0, s, qkv, ld, q_out, kc, vc, cosT, sinT, pos, slot, B,
                       S, Hq, Hkv, D, max_ctx);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_gather_rows(const int* kind, const int* idx, int rows, int cols, const bf16_t* table, const bf16_t* visual,
                          bf16_t* out, hipStream_t s) {
    VZ_CHECK_ARG(rows > 0 && cols % 8 == 0, "gather: bad shape");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows_grid(rows)), dim3(256), 0, s, kind, idx, rows, cols, table, visual, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_embed_tokens(const int* ids, int rows, int cols, const bf16_t* table, bf16_t* out, hipStream_t s) {
    return vz_launch_gather_rows(nullptr, ids, rows, cols, table, nullptr, out, s);
}

int vz_launch_copy_rows(const bf16_t* src, long src_stride, bf16_t* dst, long dst_stride, int rows, int cols, hipStream_t s) {
    VZ_CHECK_ARG(rows > 0 && cols % 8 == 0 && src_stride % 8 == 0 && dst_stride % 8 == 0, "copy_rows: bad shape");
    hipLaunchKernelGGL(copy_rows_kernel, dim3(rows_grid(rows)), dim3(256), 0, s, src, src_stride, dst, dst_stride, rows, cols);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_kv_move_rows(bf16_t* kv, size_t layer_elems, int n_layers, int max_batch, int Hkv, int max_ctx, int D, const KvMoves& mv,
                           hipStream_t s) {
    VZ_CHECK_ARG(kv && mv.n >= 1 && mv.n <= 16 && D % 8 == 0, "kv_move_rows: bad argument");
    int longest = 0;
    for (int i = 0; i < mv.n; ++i) {
        VZ_CHECK_ARG(mv.src[i] >= 0 && mv.src[i] < max_batch && mv.dst[i] >= 0 && mv.dst[i] < max_batch && mv.src[i] != mv.dst[i] &&
                         mv.len[i] >= 1 && mv.len[i] <= max_ctx, "kv_move_rows: move %d (%d -> %d, %d tokens) outside the cache", i, mv.src[i], mv.dst[i], mv.len[i]);
        for (int j = 0; j < mv.n; ++j)       // a launch has no order between its moves: no row may be written twice or read after being written
            VZ_CHECK_ARG(i == j || (mv.dst[i] != mv.dst[j] && mv.dst[i] != mv.src[j]), "kv_move_rows: moves %d and %d overlap", i, j);
        longest = std::max(longest, mv.len[i]);
    }
    const long pieces = ((long)longest * D + 2047) / 2048;
    hipLaunchKernelGGL(kv_move_rows_kernel, dim3((unsigned)pieces, (unsigned)(n_layers * 2 * Hkv), (unsigned)mv.n), dim3(256), 0, s, kv, layer_elems,
                       max_batch, Hkv, max_ctx, D, mv);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_im2col(const bf16_t* img, int T, int image, int patch, int kpad, bf16_t* out, hipStream_t s) {
    const long total = (long)T * (image / patch) * (image / patch) * kpad;
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(im2col_kernel, dim3((int)blocks), dim3(256), 0, s, img, T, image, patch, kpad, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_clip_assemble(const bf16_t* patch_out, const bf16_t* cls, const bf16_t* pos, int T, int tokens, int C,
                            bf16_t* out, hipStream_t s) {
    hipLaunchKernelGGL(clip_assemble_kernel, dim3(rows_grid((long)T * tokens)), dim3(256), 0, s, patch_out, cls, pos, T, tokens,
                       C, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_fusion(const bf16_t* hs_base, long layer_stride, int first_layer, int groups, int per_group, int T, int tokens,
                     int C, int skip, bf16_t* out, hipStream_t s) {
    const long total = (long)T * (tokens - skip) * (groups + 1) * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fusion_kernel, dim3((int)blocks), dim3(256), 0, s, hs_base, layer_stride, first_layer, groups, per_group,
                       T, tokens, C, skip, out);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_argmax(const float* logits, int rows, int cols, int* ids, int* pos, int* slot, int* len, int* out_ids,
                     int out_stride, const int* step, int max_ctx, int rope_max, int* ring, int ring_n, hipStream_t s) {
    VZ_CHECK_ARG(logits && ids && rows > 0 && cols > 0 && (!ring || (step && ring_n > 0)), "argmax: bad argument");
    hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(1024), 0, s, logits, cols, ids, pos, slot, len, out_ids, out_stride, step, max_ctx, rope_max, ring, ring_n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_step_advance(int* step, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, step);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_repack_logits(const float* gathered, float* out, int rows, int Vp, int V, int tp, hipStream_t s) {
    long blocks = ((long)rows * V + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(repack_logits_kernel, dim3((int)blocks), dim3(256), 0, s, gathered, out, rows, Vp, V, tp);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
