// Attention kernels of the Vision-Zephyr hot path (gfx950).
//
// flash_attn_kernel<HD,KT>: softmax(scale*Q K^T + mask) V without materialising the scores, for
//   * CLIP self-attention   577x577, 16 heads x 64, no mask      hf:models/clip/modeling_clip.py:259-335
//   * Q-Former self / cross 32 x (32..1920 | 576), 8 heads x 512 torch.nn.MultiheadAttention,
//                                                                ref:vis_zephyr/model/multimodal_projector/builder.py:16-25,34-39
//   * Zephyr prefill        causal + sliding window, GQA 32q/8kv x 128
//                                                                hf:models/mistral/modeling_mistral.py:84-119,139-178
// One workgroup = 4 waves x 16 query rows.  Scores are computed TRANSPOSED (S^T = K Q^T with
// v_mfma_f32_16x16x32_bf16: keys on the accumulator rows, the query on the lane), so that
//   - the row statistics of a query live in one lane column (reduce over 4 lane groups: 2 shuffles),
//   - exp(S^T) packed to bf16 is already the B operand of O^T += V^T P^T (no LDS round trip for P),
//   - the O^T accumulators of a lane all belong to its own query: the online-softmax rescale is
//     lane-local.
// K tiles are staged row-major in LDS (16-byte reads feed the MFMA A operand directly); V tiles are
// staged TRANSPOSED ([d][key], two keys packed per 32-bit LDS write) so the V^T A operand is two
// 8-byte reads.  GQA never materialises repeat_kv: query head h reads KV head h / (Hq/Hkv).
// fp32 softmax with running max / normaliser; P is rounded to bf16 for the MFMA, the normaliser
// is summed from the unrounded exponentials (what oracle/vz_oracle.py::_attention mirrors).
//
// attn_decode_kernel + attn_decode_combine: one query token against the KV cache (HBM-bound,
// split over the context so that B*Hq*nsplit workgroups stream the cache), lengths read from
// device memory so that a captured hipGraph replays for every step.
#include <algorithm>

#include "vz_common.h"

namespace {

struct FlashParams {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    int B, Sq, Sk, Hq, Hkv;
    long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
    float scale;
    int causal, q_pos0, window;
    const int* kv_len;
    long long* stamps;      // profiling only (vz_tune_set(16, 1)): stage cycle counts of wave 0 of the longest causal workgroup
    float* part;            // SPLIT launches: [B][Hq][nsplit][Sq] x {o[HD], m, l}
    const float *rope_cos, *rope_sin;   // v2, head_dim 128: RoPE applied to the queries as they are loaded (null: q is rotated already)
    const int* rope_pos;
};

// SPLIT (Sq <= 64, no mask): blockIdx.x cuts the key tiles instead of the query rows and the workgroup leaves its
// un-normalised O, running maximum and normaliser in p.part for flash_split_combine.
template <int HD, int KT, bool SPLIT>
__global__ __launch_bounds__(256, 1) void flash_attn_kernel(FlashParams p) {
    constexpr int KS_STRIDE = HD * 2 + 16;   // bytes per K row in LDS (+16: spreads the 16 rows of a fragment read)
    constexpr int VT_STRIDE = KT * 2 + 8;    // bytes per V^T row (d) in LDS
    constexpr int NT = KT / 16;              // 16-key score tiles per KV tile
    constexpr int DS = HD / 32;              // k-steps of the QK^T contraction
    constexpr int DT = HD / 16;              // 16-wide d tiles of O^T
    constexpr int CPK = HD / 8;              // 16-byte chunks per K/V row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + KT * KS_STRIDE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q0 = SPLIT ? 0 : blockIdx.x * 64;
    const int hk = h / (p.Hq / p.Hkv);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;

    // ---- this lane's query row (B operand of S^T = K Q^T): Q[q][ds*32 + 8g .. +7] ----
    int qrow = q0 + wave * 16 + c;
    const bool q_valid = qrow < p.Sq;
    if (!q_valid) qrow = p.Sq - 1;
    const bf16_t* qp = p.q + (size_t)b * p.q_bs + (size_t)qrow * p.q_ss + (size_t)h * p.q_hs;
    bf16x8 qf[DS];
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *(const bf16x8*)(qp + ds * 32 + g * 8);
    const int qpos = p.q_pos0 + qrow;

    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    // key range needed by this workgroup's 64 query rows
    int k_end = kv_len;
    int k_begin = 0;
    if (p.causal) {
        const int last_q = min(q0 + 63, p.Sq - 1) + p.q_pos0;
        k_end = min(k_end, last_q + 1);
        if (p.window > 0) k_begin = max(0, q0 + p.q_pos0 - p.window + 1);
    }
    int t_begin = k_begin / KT, t_end = (k_end + KT - 1) / KT;
    if (SPLIT) {
        const int per = (t_end - t_begin + (int)gridDim.x - 1) / (int)gridDim.x;
        t_begin += (int)blockIdx.x * per;
        t_end = min(t_end, t_begin + per);
    }

    const bf16_t* kbase = p.k + (size_t)b * p.k_bs + (size_t)hk * p.k_hs;
    const bf16_t* vbase = p.v + (size_t)b * p.v_bs + (size_t)hk * p.v_hs;
    const int last_key = kv_len > 0 ? kv_len - 1 : 0;

    for (int t = t_begin; t < t_end; ++t) {
        const int key0 = t * KT;
        __syncthreads();  // every wave has finished reading the previous tile
        // ---- stage K row-major ----
#pragma unroll
        for (int i = 0; i < (KT * CPK) / 256; ++i) {
            const int ch = i * 256 + tid;
            const int kr = ch / CPK, dc = ch % CPK;
            const int krow = min(key0 + kr, last_key);
            const uint4 val = *(const uint4*)(kbase + (size_t)krow * p.k_ss + dc * 8);
            *(uint4*)(Ks + kr * KS_STRIDE + dc * 16) = val;
        }
        // ---- stage V transposed: unit = (key pair, 8-wide d chunk) -> 8 packed 32-bit LDS writes ----
#pragma unroll
        for (int i = 0; i < ((KT / 2) * CPK) / 256; ++i) {
            const int u = i * 256 + tid;
            const int kp = u % (KT / 2), dc = u / (KT / 2);
            const int r0 = min(key0 + 2 * kp, last_key), r1 = min(key0 + 2 * kp + 1, last_key);
            const u16x8 a = *(const u16x8*)(vbase + (size_t)r0 * p.v_ss + dc * 8);
            const u16x8 bb = *(const u16x8*)(vbase + (size_t)r1 * p.v_ss + dc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *(unsigned*)(Vt + (dc * 8 + j) * VT_STRIDE + kp * 4) = (unsigned)a[j] | ((unsigned)bb[j] << 16);
        }
        __syncthreads();

        // ---- S^T tiles: sacc[nt][r] = S[key = key0 + nt*16 + 4g + r][q = this lane's query] ----
        f32x4 sacc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            sacc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                const bf16x8 kf = *(const bf16x8*)(Ks + (nt * 16 + c) * KS_STRIDE + ds * 64 + g * 16);
                sacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ds], sacc[nt], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (per lane column) ----
        float m_tile = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = key0 + nt * 16 + 4 * g + r;
                bool ok = kidx < kv_len;
                if (p.causal) ok = ok && kidx <= qpos && (p.window <= 0 || kidx > qpos - p.window);
                const float s = ok ? sacc[nt][r] * p.scale : -INFINITY;
                sacc[nt][r] = s;
                m_tile = fmaxf(m_tile, s);
            }
        m_tile = rows_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - m_safe);  // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(sacc[nt][r] - m_safe);
                sacc[nt][r] = e;
                psum += e;
            }
        psum = rows_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha;
        }
        // ---- O^T += V^T P^T.  k-slot j of lane group g in k-step s is key  (2s + (j>>2))*16 + 4g + (j&3) ----
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
            bf16x8 pf;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pf[r] = (__bf16)sacc[2 * s][r];
                pf[4 + r] = (__bf16)sacc[2 * s + 1][r];
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* vrow = Vt + (dt * 16 + c) * VT_STRIDE + g * 8;
                const bf16x4 v0 = *(const bf16x4*)(vrow + (2 * s) * 32);
                const bf16x4 v1 = *(const bf16x4*)(vrow + (2 * s + 1) * 32);
                const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, oacc[dt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: oacc[dt][r] = O[q = this lane's query][d = dt*16 + 4g + r] ----
    if (SPLIT) {
        if (q_valid) {
            float* pp = p.part + ((((size_t)b * p.Hq + h) * gridDim.x + blockIdx.x) * p.Sq + qrow) * (HD + 4);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *(f32x4*)(pp + dt * 16 + 4 * g) = oacc[dt];
            if (g == 0) { pp[HD] = m_run; pp[HD + 1] = l_run; }
        }
        return;
    }
    if (q_valid) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* op = p.o + (size_t)b * p.o_bs + (size_t)qrow * p.o_ss + (size_t)h * p.o_hs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 pk;
            pk.x = pack_bf16x2(oacc[dt][0] * inv, oacc[dt][1] * inv);
            pk.y = pack_bf16x2(oacc[dt][2] * inv, oacc[dt][3] * inv);
            *(uint2*)(op + dt * 16 + 4 * g) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 (head_dim 64 / 128): 4 waves x 32 query rows = 128 rows per workgroup, 64-key tiles.
//   * every K fragment read from LDS feeds two query tiles (halves the LDS bytes per MFMA of v1);
//   * V is staged ROW-major like K (plain 16-byte LDS writes) and consumed through the hardware transposing
//     read ds_read_b64_tr_b16: for the V^T A-operand of O^T += V^T P^T lane (g = lane>>4, c = lane&15) needs
//     V[key = tile*16 + 4g + r][d = d0 + c], r = 0..3, which is exactly what one tr-read returns when lane
//     4q+p of a 16-lane group supplies &V[block row q][d0 + 4p]; row stride 2*HD+32 B makes the 32-lane
//     halves conflict-free;
//   * K/V tiles are double-buffered with the split register staging (global loads for tile t+1 issued before
//     the MFMAs of tile t, LDS writes after them, one barrier per tile).
// Arithmetic, masking and rounding are identical to flash_attn_kernel (same tile order, same MFMA order).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x4 lds_tr16(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// The row statistics of a query live in lanes c, c+16, c+32, c+48: rows_max / rows_sum (vz_common.h) all-reduce them on the VALU.
// Online-softmax step of one 16-query tile.  FULL = every key of the tile is visible to every query of the wave
// (no padding, not on the causal diagonal, inside the window): the mask arithmetic is skipped - wave-uniform choice,
// identical values.  exp(x) is evaluated as exp2(x * log2e) on pre-scaled scores: one FMA + v_exp_f32 per element.
template <int NT, int DT, bool FULL>
__device__ __forceinline__ void softmax_tile(f32x4 (&sacc)[NT], float& m_run, float& l_run, f32x4 (&oacc)[DT], bf16x8 (&pf)[NT / 2],
                                             int key0, int g, int kv_len, int causal, int qpos, int window, float scale_log2) {
    float m_tile = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sv = sacc[nt][r] * scale_log2;
            if (!FULL) {
                const int kidx = key0 + nt * 16 + 4 * g + r;
                bool ok = kidx < kv_len;
                if (causal) ok = ok && kidx <= qpos && (window <= 0 || kidx > qpos - window);
                sv = ok ? sv : -INFINITY;
            }
            sacc[nt][r] = sv;
            m_tile = fmaxf(m_tile, sv);
        }
    m_tile = rows_max(m_tile);
    const float m_new = fmaxf(m_run, m_tile);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(sacc[nt][r] - m_safe);
            sacc[nt][r] = e;
            psum += e;
        }
    psum = rows_sum(psum);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {      // the running maximum did not move for any query of the wave: nothing to rescale
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha;
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
        bf16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[r] = (__bf16)sacc[2 * s2][r];
            t[4 + r] = (__bf16)sacc[2 * s2 + 1][r];
        }
        pf[s2] = t;
    }
}

__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// Deferred-maximum form of the same step (guide section 5.5 T13; production since round 2): (a) the scores stay RAW in the accumulators and
// the scale rides in the exponent's FMA - exp2(fma(s, scale * log2e, -m)); (b) the running maximum only advances when a tile's maximum
// exceeds it by more than 8 (a factor 256 in P: harmless in the fp32 sums and in bf16's exponent), so the rescale of the 8 x 4 output
// accumulators (16 packed multiplies + as many register copies per call in hipcc's code) and its exp2 leave almost every tile.  The
// decision is per QUERY - the four lanes holding a query's statistics see the same all-reduced maximum - and every term of l and of
// O^T added after a decision uses the same reference, so O / l is exact; probabilities are rounded to bf16 once, before PV, as before.
template <int NT, int DT, bool FULL>
__device__ __forceinline__ void softmax_tile_defer(f32x4 (&sacc)[NT], float& m_run, float& l_run, f32x4 (&oacc)[DT], bf16x8 (&pf)[NT / 2],
                                                   int key0, int g, int kv_len, int causal, int qpos, int window, float scale_log2) {
    // The kernel is bound by the ONE wave's instruction issue (the causal pairing leaves most SIMDs a single wave: ~675 instructions
    // per 64-key tile at >= 4 cycles each against 64 MFMAs = 1024 matrix cycles), so the common path is kept to the minimum:
    // * the tile maximum is taken per LANE (three-operand maxima) and only compared with the running one; the cross-lane all-reduce and
    //   the rescale run when some query of the wave would grow by more than 8 - rare after the first tiles;
    // * exponent arguments and row sums go through packed fp32 pairs (v_pk_fma_f32 / v_pk_add_f32);
    // * l_run is a per-lane PARTIAL sum (the four lanes of a query hold disjoint keys); the kernel's epilogue adds them once.
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    if (!FULL) {
        // branch-free visibility: kidx in [lo, hi] (the short-circuit form compiled to a chain of exec-mask branches per element)
        const int hi = causal ? min(kv_len - 1, qpos) : kv_len - 1;
        const int lo = (causal && window > 0) ? qpos - window + 1 : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kidx = key0 + nt * 16 + 4 * g + r;
                const bool ok = (kidx >= lo) & (kidx <= hi);
                sacc[nt][r] = ok ? sacc[nt][r] : -INFINITY;
            }
    }
    // per-lane maximum by three-operand maxima straight on the accumulators (fmaxf would first canonicalise every MFMA result)
    float m_lane = vmax3(sacc[0][0], sacc[0][1], sacc[0][2]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        if (nt == 0) m_lane = vmax3(m_lane, sacc[0][3], sacc[NT - 1][3]);
        else m_lane = vmax3(m_lane, sacc[nt][0], sacc[nt][1]);
        if (nt > 0 && nt < NT - 1) m_lane = vmax3(m_lane, sacc[nt][2], sacc[nt][3]);
        if (nt > 0 && nt == NT - 1) m_lane = vmax3(m_lane, sacc[nt][2], sacc[nt][2]);
    }
    if (__any(m_lane * scale_log2 > m_run + 8.0f)) {       // (m_run = -inf: any finite maximum grows it)
        const float m_tile = rows_max(m_lane) * scale_log2;  // scale > 0: the maximum commutes with it; -inf stays -inf
        // the rescale is a per-LANE conditional (the four lanes of a query agree): exec-masked in-place multiplies, so the common
        // path carries the 64 accumulator registers through untouched (a wave-uniform select of alpha made hipcc copy them all)
        if (m_tile > m_run + 8.0f) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_tile);      // m_tile is finite here; m_run = -inf gives 0
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha;
            }
            m_run = m_tile;
        }
    }
    const float m_safe = m_run == -INFINITY ? 0.f : m_run;
    const f32x2 sc2 = {scale_log2, scale_log2}, nm2 = {-m_safe, -m_safe};
    f32x2 ps = {0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x2 a0 = (f32x2){sacc[nt][0], sacc[nt][1]} * sc2 + nm2, a1 = (f32x2){sacc[nt][2], sacc[nt][3]} * sc2 + nm2;
        const f32x2 e0 = {__builtin_amdgcn_exp2f(a0[0]), __builtin_amdgcn_exp2f(a0[1])};
        const f32x2 e1 = {__builtin_amdgcn_exp2f(a1[0]), __builtin_amdgcn_exp2f(a1[1])};
        sacc[nt] = (f32x4){e0[0], e0[1], e1[0], e1[1]};
        ps += e0 + e1;
    }
    l_run += ps[0] + ps[1];
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
        bf16x8 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[r] = (__bf16)sacc[2 * s2][r];
            t[4 + r] = (__bf16)sacc[2 * s2 + 1][r];
        }
        pf[s2] = t;
    }
}

// RoPE (rotate-half) on a query row's fragments as the attention loads them: lane (c, g) holds d = ds * 32 + 8 g + j of row c, so the
// partner d +- 64 is fragment ds +- 2 of the SAME lane.  The operation order is rope_kv_kernel's as hipcc compiles it (t = partner * sin
// rounded, then one fused multiply-add), so the rotated bf16 values are the ones that kernel writes: bit-identical attention.
__device__ __forceinline__ void rope_q_frags(bf16x8 (&qf)[4], const float* __restrict__ cosT, const float* __restrict__ sinT, int pos, int g) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
        const float* cp = cosT + (size_t)pos * 64 + ds * 32 + g * 8;
        const float* sp = sinT + (size_t)pos * 64 + ds * 32 + g * 8;
        const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4), s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
        bf16x8 lo = qf[ds], hi = qf[ds + 2];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float cj = j < 4 ? c0[j] : c1[j - 4], sj = j < 4 ? s0[j] : s1[j - 4];
            const float x = (float)lo[j], y = (float)hi[j];
            float t_lo = y * sj, t_hi = x * sj;
            asm volatile("" : "+v"(t_lo), "+v"(t_hi));          // the products are rounded on their own (no contraction into the sums below)
            lo[j] = (__bf16)__builtin_fmaf(x, cj, -t_lo);
            hi[j] = (__bf16)__builtin_fmaf(y, cj, t_hi);
        }
        qf[ds] = lo; qf[ds + 2] = hi;
    }
}

template <int HD, bool STAMP, bool DEFER>
__global__ __launch_bounds__(256, 2) void flash_attn2_kernel(FlashParams p) {
    constexpr int KT = 64;
    // K arrives by LDS-DMA (global_load_lds, 16 bytes per lane, no staging registers, no ds_write): rows of HD * 2 bytes with NO
    // padding - the DMA image is lane-linear, 1 KiB per wave-instruction - and an XOR swizzle of the 16-byte chunk index with the row,
    // applied on the SOURCE address and again on the fragment read (16 lanes of a fragment read = 16 rows, same logical chunk ->
    // 16 distinct physical chunks: conflict-free).  head_dim 128: 16 chunks per row, chunk ^= row & 15; head_dim 64: 8 chunks per
    // row and two rows per 256-byte bank sweep, chunk ^= (row >> 1) & 7.
    constexpr int KS = HD * 2;               // K row stride in LDS (bytes)
    // V arrives the same way (round 2; it was staged through 16 VGPRs + ds_write_b128 before): rows of HD * 2 bytes, consumed by the
    // transposing read ds_read_b64_tr_b16 - lane (g, c) takes 8 bytes of row 4g + (c >> 2) at column pair-of-chunks dt, bytes (c & 3) * 8.
    // One pass of that read covers 8 rows x 32 bytes, so the swizzle works on 32-byte PAIRS of chunks: pair ^= row & 7 (head_dim 128,
    // 8 pairs per row) or pair ^= (row >> 1) & 3 (head_dim 64: 4 pairs per row, two rows per 256-byte bank sweep).
    constexpr int VS = HD * 2;
    constexpr int BUF = KT * (KS + VS);
    constexpr int NT = KT / 16, DS = HD / 32, DT = HD / 16, CPK = HD / 8;
    constexpr int NLD = (KT * CPK) / 256;    // 16-byte chunks per thread per operand tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    // Causal work grows with the query block index.  Workgroups n and n + 256 tend to share a CU, so every other residency
    // round (256 workgroups = 256 / blocks-per-row rows) walks the query blocks backwards: a CU then pairs a long block with
    // a short one (speed only; any placement computes the same values).
    // The direction is a function of the (head, batch) ROW alone - every row is a bijection of its query blocks whatever the
    // block count (a parity taken from the linear workgroup id flipped in the middle of a row whenever 256 was not a multiple
    // of the blocks per row: some query blocks were computed twice and others never, for every prompt length that is not
    // a multiple of 2048 - found by tests/test_stages_gpu.py::test_full_size_request_properties).
    int qb = blockIdx.x;
    {
        const int row = blockIdx.y + gridDim.y * blockIdx.z;
        const int rows_per_round = gridDim.x >= 256 ? 1 : 256 / gridDim.x;
        if (p.causal && ((row / rows_per_round) & 1)) qb = gridDim.x - 1 - qb;
    }
    const int q0 = qb * 128;
    const int hk = h / (p.Hq / p.Hkv);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    // stage stamps (shader cycles, s_memtime) of ONE wave: the last query block of head 0 (the longest causal workgroup)
    const bool st_on = STAMP && p.stamps && qb == (int)gridDim.x - 1 && h == 0 && b == 0 && wave == 0;
    long long st_t0 = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define VZ_ST(i) if (STAMP && st_on) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_t0; st_t0 = t_; }
    if (STAMP && st_on) st_t0 = (long long)__builtin_amdgcn_s_memtime();
    long long wg_t0 = 0;
    if (STAMP && p.stamps && tid == 0) wg_t0 = (long long)__builtin_amdgcn_s_memrealtime();      // 100 MHz, chip-wide: the schedule across CUs

    bf16x8 qf0[DS], qf1[DS];
    int qrow0 = q0 + wave * 32 + c, qrow1 = qrow0 + 16;
    const bool q_valid0 = qrow0 < p.Sq, q_valid1 = qrow1 < p.Sq;
    if (!q_valid0) qrow0 = p.Sq - 1;
    if (!q_valid1) qrow1 = p.Sq - 1;
    const int qpos0 = p.q_pos0 + qrow0, qpos1 = p.q_pos0 + qrow1;
    {
        const bf16_t* qp0 = p.q + (size_t)b * p.q_bs + (size_t)qrow0 * p.q_ss + (size_t)h * p.q_hs;
        const bf16_t* qp1 = p.q + (size_t)b * p.q_bs + (size_t)qrow1 * p.q_ss + (size_t)h * p.q_hs;
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) {
            qf0[ds] = *(const bf16x8*)(qp0 + ds * 32 + g * 8);
            qf1[ds] = *(const bf16x8*)(qp1 + ds * 32 + g * 8);
        }
        if constexpr (HD == 128) {
            if (p.rope_cos) {
                rope_q_frags(qf0, p.rope_cos, p.rope_sin, p.rope_pos[(size_t)b * p.Sq + qrow0], g);
                rope_q_frags(qf1, p.rope_cos, p.rope_sin, p.rope_pos[(size_t)b * p.Sq + qrow1], g);
            }
        }
    }
    f32x4 oacc0[DT], oacc1[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { oacc0[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; oacc1[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    float m_run0 = -INFINITY, m_run1 = -INFINITY, l_run0 = 0.f, l_run1 = 0.f;

    int k_end = kv_len, k_begin = 0;
    if (p.causal) {
        const int last_q = min(q0 + 127, p.Sq - 1) + p.q_pos0;
        k_end = min(k_end, last_q + 1);
        if (p.window > 0) k_begin = max(0, q0 + p.q_pos0 - p.window + 1);
    }
    const int t_begin = k_begin / KT, t_end = (k_end + KT - 1) / KT;
    // wave-uniform bases, pinned into SGPRs (readfirstlane): the loads below then take the saddr + 32-bit voffset form, one VGPR per address
    auto uniform_ptr = [](const bf16_t* ptr) {
        const unsigned long long v = (unsigned long long)ptr;
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        return (const char*)(((unsigned long long)hi << 32) | (unsigned long long)lo);       // (the builtin returns int: widen as UNSIGNED)
    };
    const char* kbase = uniform_ptr(p.k + (size_t)b * p.k_bs + (size_t)hk * p.k_hs);
    const char* vbase = uniform_ptr(p.v + (size_t)b * p.v_bs + (size_t)hk * p.v_hs);
    const int last_key = kv_len > 0 ? kv_len - 1 : 0;

    // K: LDS-DMA; V: split register staging (macros, not lambdas: arrays captured by reference end up in scratch).  Source offsets
    // are 32-bit (one batch's K/V of one head group stays below 4 GB) on the uniform bases: one v_mad per load instead of a 64-bit chain.
    const unsigned k_ssb = (unsigned)p.k_ss * 2u, v_ssb = (unsigned)p.v_ss * 2u;
#define VZ_G_LOAD(T, BUFI)                                                                     \
    {                                                                                          \
        const int key0_ = (T) * KT;                                                            \
        char* Kd_ = smem + (BUFI) * BUF + wave * 1024;                                         \
        _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                      \
            const int ch = i * 256 + tid;                                                      \
            const int kr = ch / CPK, pc = ch % CPK;                                            \
            const int lc = CPK == 16 ? (pc ^ (kr & 15)) : (pc ^ ((kr >> 1) & 7));              \
            const int lv = CPK == 16 ? (pc ^ ((kr & 7) << 1)) : (pc ^ (((kr >> 1) & 3) << 1)); \
            const unsigned krow = (unsigned)min(key0_ + kr, last_key);                         \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + (krow * k_ssb + lc * 16)), \
                                             (__attribute__((address_space(3))) void*)(Kd_ + i * 4096), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + (krow * v_ssb + lv * 16)), \
                                             (__attribute__((address_space(3))) void*)(Kd_ + KT * KS + i * 4096), 16, 0, 0); \
        }                                                                                      \
    }
    if (t_begin < t_end) { VZ_G_LOAD(t_begin, 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // The Q fragments came from global loads.  hipcc's waitcnt pass joins the loop's paths conservatively and would
    // otherwise wait for "the loads that may still be writing qf" right after each tile's prefetch is issued, i.e.
    // drain the prefetch every iteration.  Passing the fragments through an empty asm makes their producer opaque
    // (one real wait here, none in the loop).
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) {
        asm volatile("" : "+v"(qf0[ds]));
        asm volatile("" : "+v"(qf1[ds]));
    }
    __syncthreads();
    // tr-read address of this lane inside a [16 keys][16 d] block: row q4 = (lane&15)>>2 of lane group g, columns 4p
    const int tr_off = (4 * g + (c >> 2)) * VS + (c & 3) * 8;          // + ((pair ^ tr_sw) * 32): rows 16 apart share the swizzle term
    const int tr_sw = CPK == 16 ? ((4 * g + (c >> 2)) & 7) : (((4 * g + (c >> 2)) >> 1) & 3);
    const float scale_log2 = p.scale * 1.4426950408889634f;
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
        const int key0 = t * KT;
        VZ_ST(7)
        if (t + 1 < t_end) VZ_G_LOAD(t + 1, cur ^ 1)
        VZ_ST(0)
        const char* Ks = smem + cur * BUF;
        const char* Vs = Ks + KT * KS;
        // The 64-key LDS tile is consumed as two 32-key halves, software-pipelined INSIDE the wave (the causal long / short pairing leaves
        // most CUs with one wave per SIMD for most of the kernel, so nobody else fills the gaps): the QK^T MFMAs of BOTH halves are issued
        // first, then the softmax of half 0 runs on the VALU while the matrix pipe works through half 1's QK^T, PV of half 0 is issued,
        // and the softmax of half 1 runs under it (an MFMA holds the vector issue for 8 of its 16 cycles: MI355X_MICROARCH cycle constants).
        // sched_barrier pins the order the source states.
        const int w_first = p.q_pos0 + q0 + wave * 32, w_last = w_first + 31;
        f32x4 sacc0[4], sacc1[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            sacc0[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            sacc1[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // two K fragments (8 VGPRs) in flight at a time: the kernel sits at the 256-register limit of two waves per SIMD, and a
            // spilled loop invariant costs a scratch reload behind the in-flight K/V loads (vmcnt is in-order) every tile
#pragma unroll
            for (int d2 = 0; d2 < DS; d2 += 2) {
                bf16x8 kf[2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    kf[u] = *(const bf16x8*)(Ks + (nt * 16 + c) * KS + ((((d2 + u) * 4 + g) ^ (CPK == 16 ? c : (c >> 1))) << 4));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    sacc0[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf0[d2 + u], sacc0[nt], 0, 0, 0);
                    sacc1[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], qf1[d2 + u], sacc1[nt], 0, 0, 0);
                }
            }
            if (nt == 1) __builtin_amdgcn_sched_barrier(0);
        }
        VZ_ST(1)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int hkey0 = key0 + hf * 32;
            __builtin_amdgcn_sched_barrier(0);
            f32x4 (&sa0)[2] = *(f32x4(*)[2])&sacc0[2 * hf];
            f32x4 (&sa1)[2] = *(f32x4(*)[2])&sacc1[2 * hf];
            bf16x8 pf0[1], pf1[1];
            // fully visible to this wave's 32 queries: below the causal diagonal of its first row, inside every row's
            // window, inside the valid keys (wave-uniform)
            const bool full = hkey0 + 32 <= kv_len && (!p.causal || (hkey0 + 31 <= w_first && (p.window <= 0 || hkey0 > w_last - p.window)));
            if constexpr (DEFER) {
                if (full) {
                    softmax_tile_defer<2, DT, true>(sa0, m_run0, l_run0, oacc0, pf0, hkey0, g, kv_len, p.causal, qpos0, p.window, scale_log2);
                    softmax_tile_defer<2, DT, true>(sa1, m_run1, l_run1, oacc1, pf1, hkey0, g, kv_len, p.causal, qpos1, p.window, scale_log2);
                } else {
                    softmax_tile_defer<2, DT, false>(sa0, m_run0, l_run0, oacc0, pf0, hkey0, g, kv_len, p.causal, qpos0, p.window, scale_log2);
                    softmax_tile_defer<2, DT, false>(sa1, m_run1, l_run1, oacc1, pf1, hkey0, g, kv_len, p.causal, qpos1, p.window, scale_log2);
                }
            } else if (full) {
                softmax_tile<2, DT, true>(sa0, m_run0, l_run0, oacc0, pf0, hkey0, g, kv_len, p.causal, qpos0, p.window, scale_log2);
                softmax_tile<2, DT, true>(sa1, m_run1, l_run1, oacc1, pf1, hkey0, g, kv_len, p.causal, qpos1, p.window, scale_log2);
            } else {
                softmax_tile<2, DT, false>(sa0, m_run0, l_run0, oacc0, pf0, hkey0, g, kv_len, p.causal, qpos0, p.window, scale_log2);
                softmax_tile<2, DT, false>(sa1, m_run1, l_run1, oacc1, pf1, hkey0, g, kv_len, p.causal, qpos1, p.window, scale_log2);
            }
            VZ_ST(2)
            __builtin_amdgcn_sched_barrier(0);
            // ---- O^T += V^T P^T: each V^T fragment (two transposing reads) feeds both query tiles ----
#pragma unroll
            for (int d4 = 0; d4 < DT; d4 += 2) {
                bf16x8 vf[2];      // two V^T fragments (4 transposing reads) in flight before their MFMAs
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const char* vp = Vs + tr_off + (((d4 + u) ^ tr_sw) << 5);
                    const bf16x4 v0 = lds_tr16(vp + (2 * hf) * 16 * VS);
                    const bf16x4 v1 = lds_tr16(vp + (2 * hf + 1) * 16 * VS);
                    vf[u] = (bf16x8){v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    oacc0[d4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], pf0[0], oacc0[d4 + u], 0, 0, 0);
                    oacc1[d4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], pf1[0], oacc1[d4 + u], 0, 0, 0);
                }
            }
        }
        VZ_ST(3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the K / V DMA of tile t + 1 has landed (this wave's share; the barrier covers the others')
        VZ_ST(4)
        __syncthreads();
        VZ_ST(5)
        cur ^= 1;
    }
    if (STAMP && st_on && lane == 0) {
        for (int i = 0; i < 8; ++i) p.stamps[i] = st_acc[i];
        p.stamps[8] = t_end - t_begin;
    }
    if (STAMP && p.stamps && tid == 0) {        // per workgroup: start, end (100 MHz), hardware id (XCC | SE | CU ...), tiles
        const int wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (wg < 2048) {
            long long* r = p.stamps + 16 + (size_t)wg * 4;
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            r[0] = wg_t0; r[1] = (long long)__builtin_amdgcn_s_memrealtime();
            r[2] = ((long long)(xcc & 15) << 32) | hwid; r[3] = ((long long)qb << 32) | (t_end - t_begin);
        }
    }
#undef VZ_ST
#undef VZ_G_LOAD
    if constexpr (DEFER) { l_run0 = rows_sum(l_run0); l_run1 = rows_sum(l_run1); }       // per-lane partial row sums -> the query's sum
    if (q_valid0) {
        const float inv = l_run0 > 0.f ? 1.0f / l_run0 : 0.f;
        bf16_t* op = p.o + (size_t)b * p.o_bs + (size_t)qrow0 * p.o_ss + (size_t)h * p.o_hs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 pk;
            pk.x = pack_bf16x2(oacc0[dt][0] * inv, oacc0[dt][1] * inv);
            pk.y = pack_bf16x2(oacc0[dt][2] * inv, oacc0[dt][3] * inv);
            *(uint2*)(op + dt * 16 + 4 * g) = pk;
        }
    }
    if (q_valid1) {
        const float inv = l_run1 > 0.f ? 1.0f / l_run1 : 0.f;
        bf16_t* op = p.o + (size_t)b * p.o_bs + (size_t)qrow1 * p.o_ss + (size_t)h * p.o_hs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            uint2 pk;
            pk.x = pack_bf16x2(oacc1[dt][0] * inv, oacc1[dt][1] * inv);
            pk.y = pack_bf16x2(oacc1[dt][2] * inv, oacc1[dt][3] * inv);
            *(uint2*)(op + dt * 16 + 4 * g) = pk;
        }
    }
}

// grid (Sq, Hq, B), HD / 4 threads: merge the nsplit partial softmaxes of one query row (4 output columns per thread)
template <int HD>
__global__ __launch_bounds__(HD / 4) void flash_split_combine(FlashParams p, int nsplit) {
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x * 4;
    const float* pp = p.part + ((((size_t)b * p.Hq + h) * nsplit) * p.Sq + q) * (HD + 4);
    const size_t sstride = (size_t)p.Sq * (HD + 4);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pp[s * sstride + HD]);
    float l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * sstride + HD];
        const float w = ms == -INFINITY ? 0.f : __expf(ms - m);
        l += w * pp[s * sstride + HD + 1];
        const f32x4 o = *(const f32x4*)(pp + s * sstride + d);
        acc[0] += w * o[0]; acc[1] += w * o[1]; acc[2] += w * o[2]; acc[3] += w * o[3];
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    uint2 pk;
    pk.x = pack_bf16x2(acc[0] * inv, acc[1] * inv);
    pk.y = pack_bf16x2(acc[2] * inv, acc[3] * inv);
    *(uint2*)(p.o + (size_t)b * p.o_bs + (size_t)q * p.o_ss + (size_t)h * p.o_hs + d) = pk;
}

static int g_attn_version = 3;     // 1 = v1 kernel for every head_dim (tests / A-B), 2 = v2 (head_dim 64 / 128) with the classic online-softmax step, 3 = v2 with the deferred-maximum step (production)
template <int HD>
int launch_flash2(const FlashParams& p, hipStream_t s) {
    constexpr int LDS = 2 * 64 * (HD * 2 + HD * 2);
    { int r = vz_init_attention_kernels(); if (r) return r; }
    dim3 grid((p.Sq + 127) / 128, p.Hq, p.B);
    if (p.stamps) {
        hipLaunchKernelGGL((flash_attn2_kernel<HD, true, true>), grid, dim3(256), LDS, s, p);
        VZ_LAUNCH_CHECK();
        return VZ_OK;
    }
    if (g_attn_version == 2) hipLaunchKernelGGL((flash_attn2_kernel<HD, false, false>), grid, dim3(256), LDS, s, p);
    else hipLaunchKernelGGL((flash_attn2_kernel<HD, false, true>), grid, dim3(256), LDS, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

template <int HD, int KT>
int launch_flash(const FlashParams& p, hipStream_t s) {
    constexpr int LDS = KT * (HD * 2 + 16) + HD * (KT * 2 + 8);
    { int r = vz_init_attention_kernels(); if (r) return r; }
    dim3 grid((p.Sq + 63) / 64, p.Hq, p.B);
    hipLaunchKernelGGL((flash_attn_kernel<HD, KT, false>), grid, dim3(256), LDS, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int g_attn_split = 0;              // vz_tune_set(23, v): 0 = automatic key split of few-row head_dim-512 launches, 1 = never, n >= 2 = n splits

// Few query rows against many keys (Q-Former cross-attention, 32 x 576 per tile and head): B * Hq workgroups alone leave most
// of the chip idle and each one walks its 18 key tiles serially; with a workspace the key tiles are spread over ~one workgroup per CU.
template <int HD, int KT>
int launch_flash_split(const FlashParams& p, int nsplit, hipStream_t s) {
    constexpr int LDS = KT * (HD * 2 + 16) + HD * (KT * 2 + 8);
    { int r = vz_init_attention_kernels(); if (r) return r; }
    hipLaunchKernelGGL((flash_attn_kernel<HD, KT, true>), dim3(nsplit, p.Hq, p.B), dim3(256), LDS, s, p);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL((flash_split_combine<HD>), dim3(p.Sq, p.Hq, p.B), dim3(HD / 4), 0, s, p, nsplit);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

// ------------------------------------------------------------------------------------------------
// decode attention
// ------------------------------------------------------------------------------------------------
struct DecodeParams {
    const bf16_t *q, *kc, *vc;
    bf16_t* o;
    float* part;
    int B, Hq, Hkv, max_ctx, nsplit, window;
    float scale;
    const int* ctx_len;
};

constexpr int DEC_D = 128;
constexpr int DEC_MAX_CHUNK = 512;  // keys per split handled through LDS scores

// grid (nsplit, Hq, B).  part layout per (b,h,split): [m, l, o[128]]
__global__ __launch_bounds__(256) void attn_decode_kernel(DecodeParams p) {
    __shared__ float sc[DEC_MAX_CHUNK];
    __shared__ float red[8];
    __shared__ float obuf[2][DEC_D];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int hk = h / (p.Hq / p.Hkv);
    const int ctx = p.ctx_len[b];
    int lo_vis = 0;
    if (p.window > 0) lo_vis = max(0, ctx - p.window);  // the query sits at position ctx-1
    const int span = ctx - lo_vis;
    const int chunk = (span + p.nsplit - 1) / p.nsplit;
    const int k0 = lo_vis + split * chunk;
    const int k1 = min(ctx, k0 + chunk);
    float* po = p.part + (((size_t)b * p.Hq + h) * p.nsplit + split) * (DEC_D + 2);
    if (k0 >= k1) {  // empty split (uniform over the block)
        if (tid == 0) { po[0] = -INFINITY; po[1] = 0.f; }
        if (tid < DEC_D) po[2 + tid] = 0.f;
        return;
    }
    const bf16_t* kb = p.kc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * DEC_D;
    const bf16_t* vb = p.vc + ((size_t)b * p.Hkv + hk) * (size_t)p.max_ctx * DEC_D;
    // q fragment: 16 lanes cover one key row of 128 d, 8 d each
    const int sub = lane & 15, grp = lane >> 4;
    const u16x8 qv = *(const u16x8*)(p.q + ((size_t)b * p.Hq + h) * DEC_D + sub * 8);
    float qf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[j] = bf16_to_f32(qv[j]);

    float m_run = -INFINITY, l_run = 0.f;
    float oacc = 0.f;  // thread (d = tid & 127, half = tid >> 7)
    const int d = tid & 127, half = tid >> 7;
    for (int c0 = k0; c0 < k1; c0 += DEC_MAX_CHUNK) {
        const int n = min(DEC_MAX_CHUNK, k1 - c0);
        __syncthreads();
        // scores: each 16-lane group takes one key per pass; 16 keys per block pass
        for (int kk = wave * 4 + grp; kk < n; kk += 16) {
            const u16x8 kv = *(const u16x8*)(kb + (size_t)(c0 + kk) * DEC_D + sub * 8);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += qf[j] * bf16_to_f32(kv[j]);
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            if (sub == 0) sc[kk] = s * p.scale;
        }
        __syncthreads();
        float mx = -INFINITY;
        for (int kk = tid; kk < n; kk += 256) mx = fmaxf(mx, sc[kk]);
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        const float m_new = fmaxf(m_run, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        float ps = 0.f;
        for (int kk = tid; kk < n; kk += 256) { const float e = __expf(sc[kk] - m_new); sc[kk] = e; ps += e; }
        ps = wave_sum(ps);
        if (lane == 0) red[4 + wave] = ps;
        __syncthreads();
        const float alpha = __expf(m_run - m_new);
        l_run = l_run * alpha + red[4] + red[5] + red[6] + red[7];
        m_run = m_new;
        float a = 0.f;
        for (int kk = half; kk < n; kk += 2) a += sc[kk] * bf16_to_f32(vb[(size_t)(c0 + kk) * DEC_D + d]);
        oacc = oacc * alpha + a;
    }
    obuf[half][d] = oacc;
    __syncthreads();
    if (tid < DEC_D) po[2 + tid] = obuf[0][tid] + obuf[1][tid];
    if (tid == 0) { po[0] = m_run; po[1] = l_run; }
}

// grid (Hq, B), 128 threads: merge the nsplit partials
__global__ __launch_bounds__(128) void attn_decode_combine(const float* __restrict__ part, bf16_t* __restrict__ o, int Hq, int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pp = part + ((size_t)b * Hq + h) * nsplit * (DEC_D + 2);
    float m = -INFINITY;
    for (int s = 0; s < nsplit; ++s) m = fmaxf(m, pp[s * (DEC_D + 2)]);
    float l = 0.f, acc = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * (DEC_D + 2)];
        const float w = ms == -INFINITY ? 0.f : __expf(ms - m);
        l += w * pp[s * (DEC_D + 2) + 1];
        acc += w * pp[s * (DEC_D + 2) + 2 + d];
    }
    o[((size_t)b * Hq + h) * DEC_D + d] = f32_to_bf16(l > 0.f ? acc / l : 0.f);
}

}  // namespace

template <int HD, int KT>
static int set_flash_attr() {
    constexpr int LDS = KT * (HD * 2 + 16) + HD * (KT * 2 + 8);
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn_kernel<HD, KT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (HD == 512) VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn_kernel<HD, KT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    return VZ_OK;
}

static long long* g_attn_stamps = nullptr;
int g_attn_stamp_on = 0;           // vz_tune_set(16, 1)
int vz_attn_read_stamps(long long* host16) {          // 16 stage words + 2048 x 4 per-workgroup schedule words
    VZ_CHECK_ARG(host16 && g_attn_stamps, "attn stamps: not initialised");
    VZ_CHECK_HIP(hipDeviceSynchronize());
    VZ_CHECK_HIP(hipMemcpy(host16, g_attn_stamps, (16 + 2048 * 4) * sizeof(long long), hipMemcpyDeviceToHost));
    return VZ_OK;
}
void vz_set_attn_version(int v) { g_attn_version = v; }
int vz_attn_version() { return g_attn_version; }
void vz_set_attn_split(int v) { g_attn_split = v; }

int vz_init_attention_kernels() {
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    int r;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<64, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (64 * 2 + 64 * 2)));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<64, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (64 * 2 + 64 * 2)));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<128, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (128 * 2 + 128 * 2)));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (128 * 2 + 128 * 2)));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (64 * 2 + 64 * 2)));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)flash_attn2_kernel<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * (128 * 2 + 128 * 2)));
    VZ_CHECK_HIP(hipMalloc((void**)&g_attn_stamps, (16 + 2048 * 4) * sizeof(long long)));
    VZ_CHECK_HIP(hipMemset(g_attn_stamps, 0, (16 + 2048 * 4) * sizeof(long long)));
    if ((r = set_flash_attr<64, 64>())) return r;
    if ((r = set_flash_attr<128, 64>())) return r;
    if ((r = set_flash_attr<512, 32>())) return r;
    return VZ_OK;
}

int vz_launch_attention(const AttnArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(a.q && a.k && a.v && a.o, "attention: null pointer");
    VZ_CHECK_ARG(a.B > 0 && a.Sq > 0 && a.Sk > 0 && a.Hq > 0 && a.Hkv > 0 && a.Hq % a.Hkv == 0,
                 "attention: bad shape B=%d Sq=%d Sk=%d Hq=%d Hkv=%d", a.B, a.Sq, a.Sk, a.Hq, a.Hkv);
    VZ_CHECK_ARG(a.Hq <= 65535 && a.B <= 65535, "attention: grid too large");
    const long strides[] = {a.q_bs, a.q_ss, a.q_hs, a.k_bs, a.k_ss, a.k_hs, a.v_bs, a.v_ss, a.v_hs, a.o_bs, a.o_ss, a.o_hs};
    for (long st : strides) VZ_CHECK_ARG(st % 8 == 0, "attention: strides must be multiples of 8 elements");
    VZ_CHECK_ARG(((uintptr_t)a.q & 15) == 0 && ((uintptr_t)a.k & 15) == 0 && ((uintptr_t)a.v & 15) == 0 &&
                     ((uintptr_t)a.o & 15) == 0, "attention: pointers must be 16-byte aligned");
    FlashParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.o = a.o;
    p.B = a.B; p.Sq = a.Sq; p.Sk = a.Sk; p.Hq = a.Hq; p.Hkv = a.Hkv;
    p.q_bs = a.q_bs; p.q_ss = a.q_ss; p.q_hs = a.q_hs; p.k_bs = a.k_bs; p.k_ss = a.k_ss; p.k_hs = a.k_hs;
    p.v_bs = a.v_bs; p.v_ss = a.v_ss; p.v_hs = a.v_hs; p.o_bs = a.o_bs; p.o_ss = a.o_ss; p.o_hs = a.o_hs;
    p.scale = a.scale; p.causal = a.causal; p.q_pos0 = a.q_pos0; p.window = a.window; p.kv_len = a.kv_len;
    p.stamps = g_attn_stamp_on ? g_attn_stamps : nullptr;
    p.part = nullptr;
    p.rope_cos = a.rope_cos; p.rope_sin = a.rope_sin; p.rope_pos = a.rope_pos;
    VZ_CHECK_ARG(!a.rope_cos || (a.head_dim == 128 && a.rope_sin && a.rope_pos && g_attn_version != 1 && !a.part),
                 "attention: RoPE at the query load is built into the head_dim-128 v2 kernel only");
    if (a.head_dim == 512 && a.part && g_attn_split != 1 && a.Sq <= 64 && !a.causal && !a.kv_len) {
        // the split is a function of Sk alone (three 32-key tiles per workgroup): a query row's result never depends on how many
        // tiles or samples share the launch (tests/test_stages_gpu.py::test_continuous_batching_matches_static_batches)
        const int ntiles = (a.Sk + 31) / 32;
        const int nsplit = g_attn_split >= 2 ? std::min(g_attn_split, ntiles) : (ntiles >= 6 ? (ntiles + 2) / 3 : 1);
        if (nsplit >= 2 && nsplit <= 1024 && (size_t)a.B * a.Hq * nsplit * a.Sq * (512 + 4) <= a.part_floats) {
            p.part = a.part;
            return launch_flash_split<512, 32>(p, nsplit, s);
        }
    }
    if (g_attn_version != 1) {
        if (a.head_dim == 64) return launch_flash2<64>(p, s);
        if (a.head_dim == 128) return launch_flash2<128>(p, s);
    }
    switch (a.head_dim) {
        case 64: return launch_flash<64, 64>(p, s);
        case 128: return launch_flash<128, 64>(p, s);
        case 512: return launch_flash<512, 32>(p, s);
        default: vz_set_error("attention: head_dim %d unsupported (64, 128, 512)", a.head_dim); return VZ_ERR_ARG;
    }
}

int vz_launch_attn_decode(const AttnDecodeArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(a.D == DEC_D, "attn_decode: head_dim %d unsupported (128)", a.D);
    VZ_CHECK_ARG(a.nsplit >= 1 && a.nsplit <= 64 && a.Hq % a.Hkv == 0, "attn_decode: bad split/heads");
    DecodeParams p;
    p.q = a.q; p.kc = a.kc; p.vc = a.vc; p.o = a.o; p.part = a.part;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.max_ctx = a.max_ctx; p.nsplit = a.nsplit; p.window = a.window;
    p.scale = a.scale; p.ctx_len = a.ctx_len;
    hipLaunchKernelGGL(attn_decode_kernel, dim3(a.nsplit, a.Hq, a.B), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_decode_combine, dim3(a.Hq, a.B), dim3(128), 0, s, (const float*)a.part, a.o, a.Hq, a.nsplit);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
