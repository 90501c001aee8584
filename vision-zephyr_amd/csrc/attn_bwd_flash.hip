// Tile-resident attention backward for the Stage-1 training step (gfx950): dQ, dK, dV of softmax(scale Q K^T + mask) V for head_dim 128 with
// causal + sliding-window + padding masks and grouped KV heads (hf:models/mistral/modeling_mistral.py:139-178; what the reference trains
// through FlashAttention-2: ref:vis_zephyr/train/zephyr_flash_attn_monkey_patch.py:100-124, --model_max_length 2048 ref:script/pretrain.sh:44).
// Nothing of size Sq x Sk ever reaches HBM: the probabilities are recomputed from Q and K per 64 x 64 tile, in LDS / registers.
//
// Two kernels, both of the form "16 resident rows per wave, 64-row tiles of the other side streamed through LDS":
//   flash_bwd_dq_kernel   (workgroup = 64 queries of one head): resident Q and dO rows; K / V tiles streamed TWICE -
//        pass 1: S^T = K Q^T, dP^T = V dO^T, online (m, l, t = sum e^(s-m) dP)  ->  lse = m + log2 l, delta = t / l  (= rowsum(dO o O));
//                lse / delta are also written out for the second kernel;
//        pass 2: P = exp2(s - lse), dS = P o (dP - delta) scale, dQ^T += K^T dS^T.
//   flash_bwd_dkv_kernel  (workgroup = 64 keys of one KV head): resident K and V rows; the Q / dO tiles of the group's query heads streamed
//        once: S = Q K^T, dP = dO V^T, P, dS as above, dV^T += dO^T P, dK^T += Q^T dS - the sum over the query heads of a KV head happens in
//        the accumulators (fixed order: bit-reproducible, no float atomics).
// Products are v_mfma_f32_16x16x32_bf16 in the forward kernels' arrangement (attention.hip): the streamed side on the accumulator rows, the
// resident row on the lane, so a lane's 4 x 2 packed probabilities of two 16-row sub-tiles ARE the B operand of the second product, whose
// A operand (the streamed tile transposed) comes from the row-major LDS image through ds_read_b64_tr_b16.  P and dS are rounded to bf16
// once, before their products (as the materialising path's softmax kernels round them); statistics in fp32.
// Cost: 5 + 4 tile products against the minimum of 5 (a single kernel needs float atomics for dQ); Stage-1 captions are ~200 tokens, the
// form is there for the long-sequence case where the materialised P of train_engine.inc would be gigabytes per layer.
#include "vz_common.h"

namespace {

constexpr int FB_D = 128;
constexpr int FB_T = 64;                    // rows per streamed tile / resident rows per workgroup
constexpr int FB_TS = FB_D * 2 + 32;        // LDS row stride (bytes): 16-byte row reads and the transposing reads both spread over the banks
constexpr int FB_TILE = FB_T * FB_TS;
constexpr float LOG2E = 1.4426950408889634f;

struct FlashBwdParams {
    const bf16_t *q, *k, *v, *dO;
    long q_bs, q_ss, q_hs, k_bs, k_ss, k_hs, v_bs, v_ss, v_hs, o_bs, o_ss, o_hs;
    bf16_t* dq; long dq_bs, dq_ss, dq_hs;
    void *dk, *dv; int dkv_fp32; long dk_bs, dk_ss, dk_hs;
    float *lse, *delta;                     // [B][Hq][Sq]: log2-domain logsumexp of the scaled scores, rowsum(dO o O)
    int B, Sq, Sk, Hq, Hkv;
    float scale; int causal, window;
    const int* kv_len;
};

__device__ __forceinline__ bf16x4 tr16(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// rows [row0, row0 + 64) (clamped to last_row) x 128 of a strided bf16 matrix -> registers (4 x 16 bytes per thread) -> LDS tile: the loads of
// tile t + 1 are issued before the products of tile t and land under them
struct Tile4 { uint4 a, b, c, d; };          // (named members, passed by value: an array handed around by reference ends up in scratch)
__device__ __forceinline__ uint4 tile_load1(const bf16_t* base, long row_stride, int row0, int last_row, int i, int tid) {
    const int ch = i * 256 + tid, r = ch >> 4, c16 = ch & 15;
    return *(const uint4*)(base + (size_t)min(row0 + r, last_row) * row_stride + c16 * 8);
}
__device__ __forceinline__ Tile4 tile_load(const bf16_t* base, long row_stride, int row0, int last_row, int tid) {
    Tile4 t;
    t.a = tile_load1(base, row_stride, row0, last_row, 0, tid); t.b = tile_load1(base, row_stride, row0, last_row, 1, tid);
    t.c = tile_load1(base, row_stride, row0, last_row, 2, tid); t.d = tile_load1(base, row_stride, row0, last_row, 3, tid);
    return t;
}
__device__ __forceinline__ void tile_store(const Tile4 t, char* dst, int tid) {
    const int r = tid >> 4, c16 = tid & 15;          // chunk i * 256 + tid: row i * 16 + r
    *(uint4*)(dst + (r) * FB_TS + c16 * 16) = t.a;
    *(uint4*)(dst + (16 + r) * FB_TS + c16 * 16) = t.b;
    *(uint4*)(dst + (32 + r) * FB_TS + c16 * 16) = t.c;
    *(uint4*)(dst + (48 + r) * FB_TS + c16 * 16) = t.d;
}

// Z[nt][r] = sum_d T[nt * 16 + 4 g + r][d] * R[c][d]   (T = the LDS tile, R = the lane's resident row fragments)
__device__ __forceinline__ void tile_product(const char* tile, const bf16x8 (&rf)[4], f32x4 (&z)[4], int c, int g) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        z[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) {
            const bf16x8 tf = *(const bf16x8*)(tile + (nt * 16 + c) * FB_TS + (ds * 4 + g) * 16);
            z[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf, rf[ds], z[nt], 0, 0, 0);
        }
    }
}

// acc^T[dt] += T^T[d = 16 dt + 4 g + r][rows of the tile] . pf   (pf[s2] = the lane's packed values of sub-tiles 2 s2, 2 s2 + 1)
__device__ __forceinline__ void tile_accumulate(const char* tile, const bf16x8 (&pf)[2], f32x4 (&acc)[8], int c, int g) {
    const char* tp = tile + (4 * g + (c >> 2)) * FB_TS + (c & 3) * 8;
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x4 a0 = tr16(tp + dt * 32 + (2 * s2) * 16 * FB_TS);
            const bf16x4 a1 = tr16(tp + dt * 32 + (2 * s2 + 1) * 16 * FB_TS);
            const bf16x8 af = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf[s2], acc[dt], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ bool visible(int key, int qpos, int kv_len, int causal, int window) {
    bool ok = key < kv_len;
    if (causal) ok = ok && key <= qpos && (window <= 0 || key > qpos - window);
    return ok;
}

__device__ __forceinline__ void load_rows(const bf16_t* rowp, bf16x8 (&f)[4], int g) {
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) f[ds] = *(const bf16x8*)(rowp + ds * 32 + g * 8);
}

// grid (ceil(Sq / 64), Hq, B)
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_kernel(FlashBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * FB_TILE];
    char* Ks = smem;
    char* Vs = smem + FB_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * FB_T;
    const int hk = h / (p.Hq / p.Hkv);
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const int qrow_raw = q0 + wave * 16 + c;
    const bool q_valid = qrow_raw < p.Sq;
    const int qrow = q_valid ? qrow_raw : p.Sq - 1;
    bf16x8 qf[4], dof[4];
    load_rows(p.q + (size_t)b * p.q_bs + (size_t)qrow * p.q_ss + (size_t)h * p.q_hs, qf, g);
    load_rows(p.dO + (size_t)b * p.o_bs + (size_t)qrow * p.o_ss + (size_t)h * p.o_hs, dof, g);
    int k_end = kv_len, k_begin = 0;
    if (p.causal) {
        k_end = min(k_end, min(q0 + FB_T - 1, p.Sq - 1) + 1);
        if (p.window > 0) k_begin = max(0, q0 - p.window + 1);
    }
    const int t_begin = k_begin / FB_T, t_end = (k_end + FB_T - 1) / FB_T;
    const bf16_t* kb = p.k + (size_t)b * p.k_bs + (size_t)hk * p.k_hs;
    const bf16_t* vb = p.v + (size_t)b * p.v_bs + (size_t)hk * p.v_hs;
    const float scale_log2 = p.scale * LOG2E;
    const int last_key = p.Sk - 1;

    // ---- pass 1: the row statistics ----
    float m_run = -INFINITY, l_run = 0.f, t_run = 0.f;
    Tile4 ka = tile_load(kb, p.k_ss, t_begin * FB_T, last_key, tid);       // (unconditional: rows are clamped)
    Tile4 va = tile_load(vb, p.v_ss, t_begin * FB_T, last_key, tid);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        tile_store(ka, Ks, tid);
        tile_store(va, Vs, tid);
        __syncthreads();
        {   // the next tile (after the last one of pass 1: the first one of pass 2)
            const int tn = t + 1 < t_end ? t + 1 : t_begin;
            ka = tile_load(kb, p.k_ss, tn * FB_T, last_key, tid);
            va = tile_load(vb, p.v_ss, tn * FB_T, last_key, tid);
        }
        f32x4 s[4], dp[4];
        tile_product(Ks, qf, s, c, g);
        tile_product(Vs, dof, dp, c, g);
        float m_tile = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * FB_T + nt * 16 + 4 * g + r;
                const float sv = visible(key, qrow, kv_len, p.causal, p.window) ? s[nt][r] * scale_log2 : -INFINITY;
                s[nt][r] = sv;
                m_tile = fmaxf(m_tile, sv);
            }
        m_tile = rows_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        float ps = 0.f, ts = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[nt][r] - m_safe);
                ps += e;
                ts += e != 0.f ? e * dp[nt][r] : 0.f;          // (a masked position's dP may come from rows nobody wrote)
            }
        l_run = l_run * alpha + rows_sum(ps);
        t_run = t_run * alpha + rows_sum(ts);
        m_run = m_new;
    }
    const float lse = l_run > 0.f ? m_run + __builtin_amdgcn_logf(l_run) : INFINITY;      // v_log_f32 = log2
    const float delta = l_run > 0.f ? t_run / l_run : 0.f;
    if (q_valid && g == 0) {
        const size_t o = ((size_t)b * p.Hq + h) * p.Sq + qrow;
        p.lse[o] = lse;
        p.delta[o] = delta;
    }

    // ---- pass 2: dQ ----
    f32x4 acc[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        tile_store(ka, Ks, tid);
        tile_store(va, Vs, tid);
        __syncthreads();
        {
            const int tn = min(t + 1, t_end - 1);       // (the last iteration re-requests its own tile: no branch around the arrays)
            ka = tile_load(kb, p.k_ss, tn * FB_T, last_key, tid);
            va = tile_load(vb, p.v_ss, tn * FB_T, last_key, tid);
        }
        f32x4 s[4], dp[4];
        tile_product(Ks, qf, s, c, g);
        tile_product(Vs, dof, dp, c, g);
        bf16x8 dsf[2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t * FB_T + nt * 16 + 4 * g + r;
                float dsv = 0.f;
                if (visible(key, qrow, kv_len, p.causal, p.window)) {
                    const __bf16 pb = (__bf16)__builtin_amdgcn_exp2f(s[nt][r] * scale_log2 - lse);
                    const float pv = (float)pb;
                    dsv = pv != 0.f ? pv * (dp[nt][r] - delta) * p.scale : 0.f;
                }
                dsf[nt >> 1][(nt & 1) * 4 + r] = (__bf16)dsv;
            }
        tile_accumulate(Ks, dsf, acc, c, g);
    }
    if (q_valid) {
        bf16_t* op = p.dq + (size_t)b * p.dq_bs + (size_t)qrow * p.dq_ss + (size_t)h * p.dq_hs;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            uint2 pk;
            pk.x = pack_bf16x2(acc[dt][0], acc[dt][1]);
            pk.y = pack_bf16x2(acc[dt][2], acc[dt][3]);
            *(uint2*)(op + dt * 16 + 4 * g) = pk;
        }
    }
}

// grid (ceil(Sk / 64), Hkv, B)
__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_kernel(FlashBwdParams p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * FB_TILE + 2 * FB_T * 4];
    char* Qs = smem;
    char* Os = smem + FB_TILE;
    float* lse_s = (float*)(smem + 2 * FB_TILE);
    float* del_s = lse_s + FB_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b = blockIdx.z, hk = blockIdx.y, k0 = blockIdx.x * FB_T;
    const int grp = p.Hq / p.Hkv;
    const int kv_len = p.kv_len ? min(p.kv_len[b], p.Sk) : p.Sk;
    const int key_raw = k0 + wave * 16 + c;
    const bool k_valid = key_raw < p.Sk;
    const int key = k_valid ? key_raw : p.Sk - 1;
    bf16x8 kf[4], vf[4];
    load_rows(p.k + (size_t)b * p.k_bs + (size_t)key * p.k_ss + (size_t)hk * p.k_hs, kf, g);
    load_rows(p.v + (size_t)b * p.v_bs + (size_t)key * p.v_ss + (size_t)hk * p.v_hs, vf, g);
    // queries that can see a key of this tile
    int i_begin = 0, i_end = p.Sq;
    if (p.causal) {
        i_begin = k0;
        if (p.window > 0) i_end = min(i_end, k0 + FB_T - 1 + p.window);
    }
    if (k0 >= kv_len) i_end = i_begin;                  // a tile of padding keys: nothing flows
    const int t_begin = i_begin / FB_T, t_end = (i_end + FB_T - 1) / FB_T;
    const float scale_log2 = p.scale * LOG2E;
    const int last_q = p.Sq - 1;
    f32x4 dv[8], dk[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) { dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int n_t = max(t_end - t_begin, 0), n_it = grp * n_t, n_t1 = max(n_t, 1);
    const bf16_t* q_b = p.q + (size_t)b * p.q_bs + (size_t)(hk * grp) * p.q_hs;
    const bf16_t* o_b = p.dO + (size_t)b * p.o_bs + (size_t)(hk * grp) * p.o_hs;
#define Q_BASE(it) (q_b + (size_t)((it) / n_t1) * p.q_hs)
#define O_BASE(it) (o_b + (size_t)((it) / n_t1) * p.o_hs)
    Tile4 qa = tile_load(Q_BASE(0), p.q_ss, t_begin * FB_T, last_q, tid);
    Tile4 oa = tile_load(O_BASE(0), p.o_ss, t_begin * FB_T, last_q, tid);
    for (int it = 0; it < n_it; ++it) {          // (query head of the group, query tile)
        {
            const int h = hk * grp + it / n_t1, t = t_begin + it % n_t1;
            const size_t so = ((size_t)b * p.Hq + h) * p.Sq;
            __syncthreads();
            tile_store(qa, Qs, tid);
            tile_store(oa, Os, tid);
            if (tid < FB_T) {
                const int i = t * FB_T + tid;
                lse_s[tid] = i < p.Sq ? p.lse[so + i] : INFINITY;        // a row past the end: P = exp2(-inf) = 0
                del_s[tid] = i < p.Sq ? p.delta[so + i] : 0.f;
            }
            __syncthreads();
            {
                const int in = min(it + 1, n_it - 1), tn = t_begin + in % n_t1;
                qa = tile_load(Q_BASE(in), p.q_ss, tn * FB_T, last_q, tid);
                oa = tile_load(O_BASE(in), p.o_ss, tn * FB_T, last_q, tid);
            }
            f32x4 s[4], dp[4];
            tile_product(Qs, kf, s, c, g);
            tile_product(Os, vf, dp, c, g);
            bf16x8 pf[2], dsf[2];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4 l4 = *(const f32x4*)(lse_s + nt * 16 + 4 * g), d4 = *(const f32x4*)(del_s + nt * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = t * FB_T + nt * 16 + 4 * g + r;
                    __bf16 pb = (__bf16)0.f;
                    float dsv = 0.f;
                    if (i < p.Sq && visible(key_raw, i, kv_len, p.causal, p.window)) {
                        pb = (__bf16)__builtin_amdgcn_exp2f(s[nt][r] * scale_log2 - l4[r]);
                        const float pv = (float)pb;
                        dsv = pv != 0.f ? pv * (dp[nt][r] - d4[r]) * p.scale : 0.f;
                    }
                    pf[nt >> 1][(nt & 1) * 4 + r] = pb;
                    dsf[nt >> 1][(nt & 1) * 4 + r] = (__bf16)dsv;
                }
            }
            tile_accumulate(Os, pf, dv, c, g);
            tile_accumulate(Qs, dsf, dk, c, g);
        }
    }
#undef Q_BASE
#undef O_BASE
    if (k_valid) {
        const size_t o = (size_t)b * p.dk_bs + (size_t)hk * p.dk_hs + (size_t)key * p.dk_ss;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            if (p.dkv_fp32) {
                *(f32x4*)((float*)p.dk + o + dt * 16 + 4 * g) = dk[dt];
                *(f32x4*)((float*)p.dv + o + dt * 16 + 4 * g) = dv[dt];
            } else {
                uint2 a, v;
                a.x = pack_bf16x2(dk[dt][0], dk[dt][1]); a.y = pack_bf16x2(dk[dt][2], dk[dt][3]);
                v.x = pack_bf16x2(dv[dt][0], dv[dt][1]); v.y = pack_bf16x2(dv[dt][2], dv[dt][3]);
                *(uint2*)((bf16_t*)p.dk + o + dt * 16 + 4 * g) = a;
                *(uint2*)((bf16_t*)p.dv + o + dt * 16 + 4 * g) = v;
            }
        }
    }
}

}  // namespace

size_t vz_flash_bwd_scratch_bytes(int B, int Sq, int Hq) { return (size_t)2 * B * Hq * Sq * sizeof(float) + 256; }

bool vz_flash_bwd_ok(const FlashBwdArgs& a) {
    auto al = [](long v) { return (v & 7) == 0; };
    return a.D == FB_D && a.Hq % a.Hkv == 0 && a.Sq >= 1 && a.Sk >= 1 && al(a.q_bs) && al(a.q_ss) && al(a.q_hs) && al(a.k_bs) && al(a.k_ss) && al(a.k_hs) &&
           al(a.v_bs) && al(a.v_ss) && al(a.v_hs) && al(a.o_bs) && al(a.o_ss) && al(a.o_hs) && al(a.dq_bs) && al(a.dq_ss) && al(a.dq_hs) &&
           (a.dk_bs & 3) == 0 && (a.dk_ss & 3) == 0 && (a.dk_hs & 3) == 0 &&
           (((uintptr_t)a.q | (uintptr_t)a.k | (uintptr_t)a.v | (uintptr_t)a.dO | (uintptr_t)a.dq | (uintptr_t)a.dk | (uintptr_t)a.dv) & 15) == 0;
}

int vz_launch_flash_bwd(const FlashBwdArgs& a, void* scratch, size_t scratch_bytes, hipStream_t s) {
    VZ_CHECK_ARG(a.q && a.k && a.v && a.dO && a.dq && a.dk && a.dv && scratch, "flash_bwd: null argument");
    VZ_CHECK_ARG(vz_flash_bwd_ok(a), "flash_bwd: head_dim 128, strides in multiples of 8 elements, 16-byte aligned tensors");
    VZ_CHECK_ARG(scratch_bytes >= vz_flash_bwd_scratch_bytes(a.B, a.Sq, a.Hq), "flash_bwd: scratch too small");
    FlashBwdParams p;
    p.q = a.q; p.k = a.k; p.v = a.v; p.dO = a.dO;
    p.q_bs = a.q_bs; p.q_ss = a.q_ss; p.q_hs = a.q_hs; p.k_bs = a.k_bs; p.k_ss = a.k_ss; p.k_hs = a.k_hs;
    p.v_bs = a.v_bs; p.v_ss = a.v_ss; p.v_hs = a.v_hs; p.o_bs = a.o_bs; p.o_ss = a.o_ss; p.o_hs = a.o_hs;
    p.dq = a.dq; p.dq_bs = a.dq_bs; p.dq_ss = a.dq_ss; p.dq_hs = a.dq_hs;
    p.dk = a.dk; p.dv = a.dv; p.dkv_fp32 = a.dkv_fp32; p.dk_bs = a.dk_bs; p.dk_ss = a.dk_ss; p.dk_hs = a.dk_hs;
    p.lse = (float*)(((uintptr_t)scratch + 127) & ~(uintptr_t)127);
    p.delta = p.lse + (size_t)a.B * a.Hq * a.Sq;
    p.B = a.B; p.Sq = a.Sq; p.Sk = a.Sk; p.Hq = a.Hq; p.Hkv = a.Hkv; p.scale = a.scale; p.causal = a.causal; p.window = a.window; p.kv_len = a.kv_len;
    hipLaunchKernelGGL(flash_bwd_dq_kernel, dim3((a.Sq + FB_T - 1) / FB_T, a.Hq, a.B), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    hipLaunchKernelGGL(flash_bwd_dkv_kernel, dim3((a.Sk + FB_T - 1) / FB_T, a.Hkv, a.B), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
