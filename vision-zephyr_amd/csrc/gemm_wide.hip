// Weight-streaming MFMA GEMM for 17..64 activation rows (wide decode batches): C[M <= 64, N'] = epi(x[M,K] . W[N,K]^T), gfx950.
//
// A 64-row decode step still reads every weight once, so the bound is the HBM stream (436 MB per Zephyr layer = 69 us at
// 6.3 TB/s); what the two earlier routes lose is the ACTIVATION side:
//   * gemm_skinny.hip's wide kernel gives one 16-row weight group to a workgroup whose waves split K: every workgroup pulls all
//     64 x K activations from L2 - 4 bytes of x per byte of W - and falls to 2.7 TB/s of weights at 64 rows;
//   * the 128^2 tile GEMM (gemm.hip, split-K) pads the rows to 128, keeps one 32-KiB stage in flight per workgroup and needs a
//     finalize launch per projection (gate|up 58 us = 4.0 TB/s, + 5-6 us each).
// Here a workgroup owns EIGHT weight row groups (128 rows; SwiGLU: 4 gate + 4 up groups = 64 outputs) over a K slice:
//   * wave w streams group w's fragment-tiled weights (vz_launch_tile_weights: 1 KiB contiguous per wave-instruction) straight
//     into registers, 8 steps (16 KiB) in flight per wave, refilled slot by slot across chunk boundaries;
//   * the activations of a 512-k chunk (64 rows x 512 k = 64 KiB) are staged ONCE per workgroup into LDS in B-fragment order
//     (every ds_read_b128 of a fragment is 1 KiB contiguous: conflict-free) and feed all eight waves - a quarter of a byte of x
//     from L2 per byte of W; double-buffered, one barrier per chunk;
//   * K can be split over P workgroups where N / 128 row blocks cannot fill the chip.  The P partial tiles meet WITHOUT a
//     finalize launch and without a device-side wait: write-through (sc1) partials, drained, one relaxed agent-scope ticket per
//     row block, the LAST arriver sums the P partials in split order (deterministic) and runs the epilogue (CDNA4 guide section
//     6 G16: "sc1 payload + drained ticket, last arriver told by the value its add returned").
// Measured at 64 rows inside a 32-layer step (tools/prof_batched.sh, profiles/r02_rows.txt): gate|up (224 workgroups, no split) 52.7 us
// against 57.1 + 5.2 (tile GEMM + finalize), lm_head 55 against 79 + 12; the split shapes do NOT win - down (P = 7) 38.5 against
// 28.9 + 6.2, O (P = 8 / 4) 22.9 against 14.2 + 6.2, QKV (P = 4) 22.2 against 17.0 + 4.9 - so the engine takes this kernel where
// pick_splits() = 1 and keeps the tile GEMM elsewhere (vz_wide_engine_ok).  The stream alone (MFMAs and LDS reads removed) ran
// at the same rate: what bounds it is the load side of one 8-wave workgroup per CU, not the compute.
// k assignment inside a 64-k step is gemm_skinny.hip's (lane (r, g): k = 16 g .. 16 g + 7 | + 8 .. + 15), so the tiled copies are shared.
// The split-K workspace (partials + tickets) exists per (device, stream) (vz_stream_ws, gemm.hip): launches that share it are
// stream-ordered by construction.
#include <algorithm>

#include "vz_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int CH = 8;                      // 64-k steps per activation chunk (512 k)
constexpr int NWV = 8;                     // waves per workgroup = weight row groups per workgroup

struct WideParams {
    const bf16_t* A; const bf16_t* Wt; void* C;
    const unsigned char* W8t; const float* wscale;      // e4m3 instantiation: fragment-tiled e4m3 copy (tile_weights_fp8_kernel) + one 2^e per weight row
    const float* bias; const bf16_t* residual;
    int M, N, K, lda, ldc, ldr;
    int act, out_fp32;
    int P, cps;                            // K splits; chunks per split (K / 64 == P * cps * CH)
    float* part; unsigned* ticket;         // P > 1: partial tiles [rb][P][wave][MH][64] f32x4, one ticket per row block
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
// 8-byte write-through store / L2-bypassing load at agent scope (global_store / load_dwordx2 sc1)
__device__ __forceinline__ void st2_sc1(float* p, float a, float b) {
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store((unsigned long long*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x2 ld2_sc1(const float* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (f32x2){__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32))};
}
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == VZ_ACT_QUICK_GELU) return act_quick_gelu(v);
    if (act == VZ_ACT_GELU_ERF) return act_gelu_erf(v);
    return v;
}

// MH = 16-row activation blocks (2: up to 32 rows, 4: up to 64)
template <int MH, bool SWIGLU>
__global__ __launch_bounds__(NWV * 64) void wide_tiled_kernel(WideParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // Xs[2][CH][2][MH][64 lanes] x 16 B
    constexpr int QN = CH * 2 * MH;                                   // 1-KiB fragments per chunk
    constexpr int XL = QN / NWV;                                      // staged by each wave per chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int rb = blockIdx.x, sp = blockIdx.y;
    const int S = p.K >> 6;
    // this wave's weight row group: plain: 8 consecutive groups; SwiGLU ([16 gate | 16 up] interleaved rows): waves 0..3 the gate
    // groups of output blocks 4 rb .. 4 rb + 3, waves 4..7 their up groups
    const int G = SWIGLU ? 2 * (rb * 4 + (wave & 3)) + (wave >> 2) : rb * NWV + wave;
    const int step0 = sp * p.cps * CH;
    const bf16_t* wa = p.Wt + ((size_t)G * S + step0) * 1024 + lane * 8;

    // ---- weight ring: slot u holds step u of the current chunk (16 KiB per wave in flight), refilled with the next chunk's step u right
    //      after its MFMAs.  A second bank (32 KiB per wave, counted waits that leave a whole chunk in flight) measured SLOWER - 56.8
    //      against 51.1 us on gate|up - as did non-temporal loads; see profiles/r02_rows.txt. ----
    const int last = p.cps - 1;
    u32x4 qa[CH][2];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        qa[u][0] = *(const u32x4*)(wa + (size_t)u * 1024);
        qa[u][1] = *(const u32x4*)(wa + (size_t)u * 1024 + 512);
    }

    // ---- activation staging.  LDS image of a chunk: fragment q = (s * 2 + j) * MH + h (1 KiB) holds x[16 h + m][64 s + 16 g + 8 j .. + 7] of
    //      lane (m, g) at 16-byte slot 16 g + (m ^ (2 s + j)) - the MFMA B operand of step s, half j, row block h is one ds_read_b128 per lane
    //      over 1 KiB (conflict-free).  Each wave-instruction of the staging loads ONE row's 512 k (1 KiB contiguous: 8 whole lines; the
    //      fragment-shaped load - 16 rows x 64 bytes - cost the weight stream another 10 % in tools/micro/stream_bench.hip) and scatters it:
    //      lane l holds k = 8 l .. 8 l + 7, i.e. s = l >> 3, g = (l >> 1) & 3, j = l & 1; the XOR spreads a row's 64 pieces over all bank
    //      groups (4-way conflicts on the write instead of 64-way).  Rows past M repeat row M - 1: their outputs are never stored. ----
    const bf16_t* xsrc[XL];
    int xdst[XL];
    {
        const int sj = ((lane >> 3) << 1) | (lane & 1), gq = (lane >> 1) & 3;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int row = wave + NWV * i;                      // 0 .. 16 MH - 1
            xsrc[i] = p.A + (size_t)min(row, p.M - 1) * p.lda + (size_t)step0 * 64 + lane * 8;
            xdst[i] = ((sj * MH + (row >> 4)) * 64 + gq * 16 + ((row & 15) ^ sj)) * 16;
        }
    }
    u32x4 xr[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) xr[i] = *(const u32x4*)xsrc[i];
#pragma unroll
    for (int i = 0; i < XL; ++i) *(u32x4*)(smem + xdst[i]) = xr[i];
    __syncthreads();

    f32x4 acc[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // The loop body is branch-free and order-pinned on purpose: hipcc counts its vmcnt waits exactly only while every load of the ring
    // is issued unconditionally (with the refills behind `if (more)` it drained the queue - the fresh L2 loads of the next chunk's
    // activations included - at the head of every chunk), and left alone its scheduler sinks every load of an iteration below the last
    // MFMA (wait for everything, compute, then load).  The last chunk re-requests its own steps / activations (L2 hits, never used).
    int buf = 0;
    for (int c = 0; c < p.cps; ++c) {
        const int cn = c < last ? c + 1 : c;
#pragma unroll
        for (int i = 0; i < XL; ++i) xr[i] = *(const u32x4*)(xsrc[i] + (size_t)cn * CH * 64);
        const char* xs = smem + (size_t)buf * QN * 1024 + (lane & 48) * 16;
        const bf16_t* wnext = wa + (size_t)cn * CH * 1024;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                const u32x4 b0 = *(const u32x4*)(xs + (size_t)((u * 2 + 0) * MH + h) * 1024 + ((fr ^ (u * 2 + 0)) << 4));
                const u32x4 b1 = *(const u32x4*)(xs + (size_t)((u * 2 + 1) * MH + h) * 1024 + ((fr ^ (u * 2 + 1)) << 4));
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(qa[u][0]), as_bf16x8(b0), acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(qa[u][1]), as_bf16x8(b1), acc[h], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            qa[u][0] = *(const u32x4*)(wnext + (size_t)u * 1024);
            qa[u][1] = *(const u32x4*)(wnext + (size_t)u * 1024 + 512);
            __builtin_amdgcn_sched_barrier(0);
        }
        char* xd = smem + (size_t)(buf ^ 1) * QN * 1024;
#pragma unroll
        for (int i = 0; i < XL; ++i) *(u32x4*)(xd + xdst[i]) = xr[i];
        __syncthreads();           // chunk c + 1 is staged; everybody is done reading chunk c
        buf ^= 1;
    }

    // ---- K splits meet: partial tiles out (write-through), one ticket per row block, the last arriver sums them in split order.
    //      Partials are stored as [tile h][half][64 lanes][2 floats]: every 8-byte store / load instruction covers 512 contiguous bytes. ----
    if (p.P > 1) {
        float* mine = p.part + ((((size_t)rb * p.P + sp) * NWV + wave) * MH) * 256 + lane * 2;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            st2_sc1(mine + h * 256, acc[h][0], acc[h][1]);
            st2_sc1(mine + h * 256 + 128, acc[h][2], acc[h][3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ unsigned last_flag;
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(p.ticket + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag = (t == (unsigned)p.P - 1) ? 1u : 0u;
            if (last_flag) __hip_atomic_store(p.ticket + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every split has arrived: ready for the next launch
        }
        __syncthreads();           // the wave that added joins this barrier after its add returned; everyone loads behind it
        if (!last_flag) return;
#pragma unroll
        for (int h = 0; h < MH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // four splits' tiles are requested before the first is added (hipcc waits for an atomic load right before its use: one
        // round trip per batch instead of one per value); the additions run in split order whatever the arrival order was
        const float* base = p.part + (((size_t)rb * p.P * NWV + wave) * MH) * 256 + lane * 2;
        const size_t sstr = (size_t)NWV * MH * 256;
        for (int s0 = 0; s0 < p.P; s0 += 4) {
            f32x2 t[4][MH][2];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float* src = base + (size_t)min(s0 + d, p.P - 1) * sstr;
#pragma unroll
                for (int h = 0; h < MH; ++h) { t[d][h][0] = ld2_sc1(src + h * 256); t[d][h][1] = ld2_sc1(src + h * 256 + 128); }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (s0 + d < p.P) {
#pragma unroll
                    for (int h = 0; h < MH; ++h) { acc[h][0] += t[d][h][0][0]; acc[h][1] += t[d][h][0][1]; acc[h][2] += t[d][h][1][0]; acc[h][3] += t[d][h][1][1]; }
                }
            }
        }
    }

    // ---- epilogue: lane (m = fr, g) of tile h holds outputs n = 4 g .. 4 g + 3 of this wave's group for row 16 h + m ----
    if (SWIGLU) {
        // up waves hand their tiles to the gate wave of the same output block through LDS (the staging buffers are free)
        float* ex = (float*)smem;
        if (wave >= 4) {
#pragma unroll
            for (int h = 0; h < MH; ++h) *(f32x4*)(ex + (((size_t)(wave - 4) * MH + h) * 64 + lane) * 4) = acc[h];
        }
        __syncthreads();
        if (wave >= 4) return;
        const int n0 = (rb * 4 + wave) * 16 + g * 4;               // output columns of block rb * 4 + wave
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            const int m = 16 * h + fr;
            if (m >= p.M) continue;
            const f32x4 up = *(const f32x4*)(ex + (((size_t)wave * MH + h) * 64 + lane) * 4);
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t[r] = act_silu(acc[h][r]) * up[r];
                if (p.residual) t[r] += bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + r]);
            }
            if (p.out_fp32) *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = (f32x4){t[0], t[1], t[2], t[3]};
            else { uint2 pk; pk.x = pack_bf16x2(t[0], t[1]); pk.y = pack_bf16x2(t[2], t[3]); *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk; }
        }
        return;
    }
    const int n0 = G * 16 + g * 4;
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        const int m = 16 * h + fr;
        if (m >= p.M) continue;
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[r] = acc[h][r];
            if (p.bias) t[r] += p.bias[n0 + r];
            t[r] = apply_act(t[r], p.act);
            if (p.residual) t[r] += bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + r]);
        }
        if (p.out_fp32) *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = (f32x4){t[0], t[1], t[2], t[3]};
        else { uint2 pk; pk.x = pack_bf16x2(t[0], t[1]); pk.y = pack_bf16x2(t[2], t[3]); *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk; }
    }
}


// 16 e4m3 weights (k = 16 g .. 16 g + 15 of one row) -> the two bf16 A fragments of a 64-k step; exact (v_cvt_scalef32_pk_bf16_fp8, scale 1)
__device__ __forceinline__ unsigned w8x2_bf16x2(unsigned w, bool hi_half) {
    return hi_half ? __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                   : __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
}
__device__ __forceinline__ void widen_w8(const u32x4 w, u32x4& lo, u32x4& hi) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    lo = (u32x4){w8x2_bf16x2(w0, false), w8x2_bf16x2(w0, true), w8x2_bf16x2(w1, false), w8x2_bf16x2(w1, true)};
    hi = (u32x4){w8x2_bf16x2(w2, false), w8x2_bf16x2(w2, true), w8x2_bf16x2(w3, false), w8x2_bf16x2(w3, true)};
}

// Fragment-tiled e4m3 weights: the 16 bytes lane (r = lane & 15, g = lane >> 4) needs for 64-k step s of row group G -
// W8[16 G + r][64 s + 16 g .. + 15], i.e. BOTH MFMA halves of the step - sit at W8t[((G * S + s) * 64 + lane) * 16], S = K / 64: one
// wave-instruction per step and group reads 1 KiB contiguous (row-major: 16 rows x 64 bytes).
__global__ __launch_bounds__(256) void tile_weights_fp8_kernel(const unsigned char* __restrict__ W8, int ldw, unsigned char* __restrict__ W8t, int N, int K) {
    const int S = K >> 6;
    const long total = (long)(N >> 4) * S * 64;           // 16-byte chunks
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long gs = i >> 6;
        const int s = (int)(gs % S), G = (int)(gs / S);
        const int r = lane & 15, g = lane >> 4;
        *(uint4*)(W8t + i * 16) = *(const uint4*)(W8 + (size_t)(16 * G + r) * ldw + 64 * s + 16 * g);
    }
}

// The W8A16 form of wide_tiled_kernel (round 3; SURVEY config 5: 17..64-row decode steps of the e4m3-weight engine): the same
// workgroup shape, LDS image of the activations, K split and epilogues; the weight stream is the e4m3 tiled copy - ONE 16-byte load per
// lane and 64-k step - widened to the two bf16 fragments in registers (exact) in front of the same bf16 MFMAs; the row's 2^e multiplies
// the finished fp32 sums (after the K splits have met).  Half the bytes per step, so the ring holds TWO chunks (16 steps = 16 KiB per wave
// in flight, the byte count the bf16 stream keeps): bank 0 = even chunks, bank 1 = odd chunks, a slot is refilled with the step two chunks
// on right after its MFMAs.  The loop handles two chunks per trip (register arrays are indexed statically), so a K split is taken only
// where it leaves an even number of chunks per workgroup (pick_splits).
template <int MH, bool SWIGLU>
__global__ __launch_bounds__(NWV * 64) void wide_tiled_fp8_kernel(WideParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];       // Xs[2][CH][2][MH][64 lanes] x 16 B
    constexpr int QN = CH * 2 * MH;
    constexpr int XL = QN / NWV;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int rb = blockIdx.x, sp = blockIdx.y;
    const int S = p.K >> 6;
    const int G = SWIGLU ? 2 * (rb * 4 + (wave & 3)) + (wave >> 2) : rb * NWV + wave;
    const int step0 = sp * p.cps * CH;
    const unsigned char* wa = p.W8t + ((size_t)G * S + step0) * 1024 + lane * 16;
    const int last = p.cps - 1;

    u32x4 q0[CH], q1[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) q0[u] = *(const u32x4*)(wa + (size_t)u * 1024);
    {
        const unsigned char* w1 = wa + (size_t)min(1, last) * CH * 1024;
#pragma unroll
        for (int u = 0; u < CH; ++u) q1[u] = *(const u32x4*)(w1 + (size_t)u * 1024);
    }

    const bf16_t* xsrc[XL];
    int xdst[XL];
    {
        const int sj = ((lane >> 3) << 1) | (lane & 1), gq = (lane >> 1) & 3;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int row = wave + NWV * i;
            xsrc[i] = p.A + (size_t)min(row, p.M - 1) * p.lda + (size_t)step0 * 64 + lane * 8;
            xdst[i] = ((sj * MH + (row >> 4)) * 64 + gq * 16 + ((row & 15) ^ sj)) * 16;
        }
    }
    u32x4 xr[XL];
#pragma unroll
    for (int i = 0; i < XL; ++i) xr[i] = *(const u32x4*)xsrc[i];
#pragma unroll
    for (int i = 0; i < XL; ++i) *(u32x4*)(smem + xdst[i]) = xr[i];
    __syncthreads();

    f32x4 acc[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int buf = 0;
    // one chunk: MFMAs of bank `q` (holding chunk c), each slot refilled with chunk min(c + 2, last)'s step; next chunk's activations staged
    auto chunk = [&](u32x4 (&q)[CH], int c) __attribute__((always_inline)) {
        const int cn = c < last ? c + 1 : c;
        const int cw = c + 2 <= last ? c + 2 : last;
#pragma unroll
        for (int i = 0; i < XL; ++i) xr[i] = *(const u32x4*)(xsrc[i] + (size_t)cn * CH * 64);
        const char* xs = smem + (size_t)buf * QN * 1024 + (lane & 48) * 16;
        const unsigned char* wnext = wa + (size_t)cw * CH * 1024;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            u32x4 lo, hi;
            widen_w8(q[u], lo, hi);
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                const u32x4 b0 = *(const u32x4*)(xs + (size_t)((u * 2 + 0) * MH + h) * 1024 + ((fr ^ (u * 2 + 0)) << 4));
                const u32x4 b1 = *(const u32x4*)(xs + (size_t)((u * 2 + 1) * MH + h) * 1024 + ((fr ^ (u * 2 + 1)) << 4));
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(lo), as_bf16x8(b0), acc[h], 0, 0, 0);
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(hi), as_bf16x8(b1), acc[h], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            q[u] = *(const u32x4*)(wnext + (size_t)u * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        char* xd = smem + (size_t)(buf ^ 1) * QN * 1024;
#pragma unroll
        for (int i = 0; i < XL; ++i) *(u32x4*)(xd + xdst[i]) = xr[i];
        __syncthreads();
        buf ^= 1;
    };
    for (int c = 0; c < p.cps; c += 2) {       // cps is even (vz_wide_ok)
        chunk(q0, c);
        chunk(q1, c + 1);
    }

    if (p.P > 1) {
        float* mine = p.part + ((((size_t)rb * p.P + sp) * NWV + wave) * MH) * 256 + lane * 2;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            st2_sc1(mine + h * 256, acc[h][0], acc[h][1]);
            st2_sc1(mine + h * 256 + 128, acc[h][2], acc[h][3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ unsigned last_flag8;
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(p.ticket + rb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_flag8 = (t == (unsigned)p.P - 1) ? 1u : 0u;
            if (last_flag8) __hip_atomic_store(p.ticket + rb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_flag8) return;
#pragma unroll
        for (int h = 0; h < MH; ++h) acc[h] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* base = p.part + (((size_t)rb * p.P * NWV + wave) * MH) * 256 + lane * 2;
        const size_t sstr = (size_t)NWV * MH * 256;
        for (int s0 = 0; s0 < p.P; s0 += 4) {
            f32x2 t[4][MH][2];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float* src = base + (size_t)min(s0 + d, p.P - 1) * sstr;
#pragma unroll
                for (int h = 0; h < MH; ++h) { t[d][h][0] = ld2_sc1(src + h * 256); t[d][h][1] = ld2_sc1(src + h * 256 + 128); }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                if (s0 + d < p.P) {
#pragma unroll
                    for (int h = 0; h < MH; ++h) { acc[h][0] += t[d][h][0][0]; acc[h][1] += t[d][h][0][1]; acc[h][2] += t[d][h][1][0]; acc[h][3] += t[d][h][1][1]; }
                }
            }
        }
    }

    // the rows' power-of-two scales, once per finished sum: lane (m, g) holds weight rows 16 G + 4 g .. + 3
    {
        const f32x4 sc = *(const f32x4*)(p.wscale + G * 16 + g * 4);
#pragma unroll
        for (int h = 0; h < MH; ++h) { acc[h][0] *= sc[0]; acc[h][1] *= sc[1]; acc[h][2] *= sc[2]; acc[h][3] *= sc[3]; }
    }

    if (SWIGLU) {
        float* ex = (float*)smem;
        if (wave >= 4) {
#pragma unroll
            for (int h = 0; h < MH; ++h) *(f32x4*)(ex + (((size_t)(wave - 4) * MH + h) * 64 + lane) * 4) = acc[h];
        }
        __syncthreads();
        if (wave >= 4) return;
        const int n0 = (rb * 4 + wave) * 16 + g * 4;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            const int m = 16 * h + fr;
            if (m >= p.M) continue;
            const f32x4 up = *(const f32x4*)(ex + (((size_t)wave * MH + h) * 64 + lane) * 4);
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                t[r] = act_silu(acc[h][r]) * up[r];
                if (p.residual) t[r] += bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + r]);
            }
            if (p.out_fp32) *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = (f32x4){t[0], t[1], t[2], t[3]};
            else { uint2 pk; pk.x = pack_bf16x2(t[0], t[1]); pk.y = pack_bf16x2(t[2], t[3]); *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk; }
        }
        return;
    }
    const int n0 = G * 16 + g * 4;
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        const int m = 16 * h + fr;
        if (m >= p.M) continue;
        float t[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[r] = acc[h][r];
            if (p.bias) t[r] += p.bias[n0 + r];
            t[r] = apply_act(t[r], p.act);
            if (p.residual) t[r] += bf16_to_f32(p.residual[(size_t)m * p.ldr + n0 + r]);
        }
        if (p.out_fp32) *(f32x4*)((float*)p.C + (size_t)m * p.ldc + n0) = (f32x4){t[0], t[1], t[2], t[3]};
        else { uint2 pk; pk.x = pack_bf16x2(t[0], t[1]); pk.y = pack_bf16x2(t[2], t[3]); *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n0) = pk; }
    }
}

constexpr size_t WIDE_PART_BYTES = (size_t)24 << 20;      // O: 32 row blocks x 8 splits x 32 KiB = 8 MiB; QKV 6; down 7
constexpr int WIDE_TICKETS = 4096;
int g_wide_cus = 256;

// K splits: the fewest residency rounds of chunk-times per CU, then the fewest splits (partials cost 2 x 32 KiB per workgroup).
// fp8: a chunk is 64 KiB of weights per CU (half the time), and a split must leave an even number of chunks (two-bank ring).
int pick_splits(int row_blocks, int chunks, bool fp8 = false) {
    int best = 1; long best_cost = -1;
    for (int P = 1; P <= chunks && P <= 32; ++P) {
        if (chunks % P) continue;
        if (fp8 && ((chunks / P) & 1)) continue;
        const long wgs = (long)row_blocks * P;
        const long rounds = (wgs + g_wide_cus - 1) / g_wide_cus;
        // one chunk-time (128 KiB of bf16 weights per CU, ~5 us) = 16 units; splits: every workgroup parks 32 KiB, the last arriver pays
        // two round trips and reads P x 32 KiB
        const long cost = rounds * (chunks / P) * (fp8 ? 8 : 16) + (P > 1 ? 14 + 4 * P : 0);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = P; }
    }
    return best;
}

}  // namespace

int g_wide_mode = 1;       // vz_tune_set(19, v): 0 = off (17..64-row decode steps keep the earlier routes)
int g_wide_fp8_splits = 0; // vz_tune_set(27, P): force the K split of the e4m3 stream (0 = pick_splits; A/B)

// what the engine's decode steps use it for: shapes that need no K split (measured faster than the tile GEMM + finalize there only)
bool vz_wide_engine_ok(const LinearArgs& a);

static int wide_splits(const LinearArgs& a) {
    const int rbs = a.N >> 7, chunks = a.K >> 9;
    const bool f8 = a.W8t != nullptr;
    if (f8 && g_wide_fp8_splits > 0 && chunks % g_wide_fp8_splits == 0 && ((chunks / g_wide_fp8_splits) & 1) == 0) return g_wide_fp8_splits;
    return pick_splits(rbs, chunks, f8);
}

bool vz_wide_ok(const LinearArgs& a) {
    if (!g_wide_mode || a.norm_w || !a.wide_ok) return false;
    if (a.W8t) { if (!a.wscale || ((uintptr_t)a.W8t & 15) != 0 || ((uintptr_t)a.wscale & 15) != 0 || ((a.K >> 9) & 1)) return false; }
    else if (!a.Wt || a.W8) return false;
    if (a.M < 17 || a.M > 64 || (a.N & 127) != 0 || (a.K & 511) != 0 || a.ldw != a.K) return false;
    if ((a.lda & 7) != 0 || (a.ldc & 3) != 0 || (a.residual && (a.ldr & 3) != 0)) return false;
    if (a.act == VZ_ACT_SWIGLU && a.bias) return false;
    if (((uintptr_t)a.bias & 15) != 0) return false;
    const int rbs = a.N >> 7;
    const int P = wide_splits(a);
    const int mh = a.M <= 32 ? 2 : 4;
    return (size_t)rbs * P * NWV * mh * 1024 <= WIDE_PART_BYTES && rbs <= WIDE_TICKETS;
}

bool vz_wide_engine_ok(const LinearArgs& a) { return vz_wide_ok(a) && (a.W8t || pick_splits(a.N >> 7, a.K >> 9) == 1); }

int vz_init_wide_kernels() {
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 2 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 2 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 4 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 4 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_fp8_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 2 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_fp8_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 2 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_fp8_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 4 * 1024));
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)wide_tiled_fp8_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH * 2 * 4 * 1024));
    int dev = 0;
    hipDeviceProp_t prop;
    VZ_CHECK_HIP(hipGetDevice(&dev));
    VZ_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (prop.multiProcessorCount > 0) g_wide_cus = prop.multiProcessorCount;
    return VZ_OK;
}

// the K-split scratch of a stream (partial tiles + zeroed tickets; per (device, stream), gemm.hip): allocated here outside a capture -
// vz_llm_decode_steps calls this for its capture stream before it captures a 17..64-row step
int vz_wide_reserve(hipStream_t s) {
    void* p = nullptr; size_t have = 0;
    { int r = vz_stream_ws(1, s, WIDE_PART_BYTES, false, &p, &have); if (r) return r; }
    { int r = vz_stream_ws(2, s, WIDE_TICKETS * sizeof(unsigned), true, &p, &have); if (r) return r; }
    return VZ_OK;
}

int vz_launch_tile_weights_fp8(const unsigned char* W8, int N, int K, int ldw, unsigned char* W8t, hipStream_t s) {
    VZ_CHECK_ARG(W8 && W8t && N > 0 && (N & 15) == 0 && K >= 64 && (K & 63) == 0 && ldw >= K && (ldw & 15) == 0 &&
                 ((uintptr_t)W8 & 15) == 0 && ((uintptr_t)W8t & 15) == 0, "tile_weights_fp8: needs N %% 16 == 0, K %% 64 == 0, 16-byte-aligned rows (N=%d K=%d)", N, K);
    const long chunks = (long)(N >> 4) * (K >> 6) * 64;
    const int blocks = (int)std::min<long>((chunks + 255) / 256, 8192);
    hipLaunchKernelGGL(tile_weights_fp8_kernel, dim3(blocks), dim3(256), 0, s, W8, ldw, W8t, N, K);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_wide(const LinearArgs& a, hipStream_t s) {
    int rc = vz_linear_check_common(a);
    if (rc) return rc;
    { int r = vz_init_wide_kernels(); if (r) return r; }
    VZ_CHECK_ARG(vz_wide_ok(a), "wide gemm: needs the tiled weight copy (bf16, or e4m3 + row scales with an even number of 512-k chunks), 17 <= M <= 64, N %% 128 == 0, K %% 512 == 0, no fused norm (M=%d N=%d K=%d)", a.M, a.N, a.K);
    WideParams p;
    p.A = a.A; p.Wt = a.Wt; p.C = a.C; p.bias = a.bias; p.residual = a.residual;
    p.W8t = a.W8t; p.wscale = a.wscale;
    p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldc = a.ldc; p.ldr = a.ldr;
    p.act = a.act; p.out_fp32 = a.out_fp32;
    const int rbs = a.N >> 7, chunks = a.K >> 9;
    p.P = wide_splits(a); p.cps = chunks / p.P;
    p.part = nullptr; p.ticket = nullptr;
    if (p.P > 1) {
        void* part = nullptr; void* tick = nullptr; size_t hp = 0, ht = 0;
        { int r = vz_stream_ws(1, s, WIDE_PART_BYTES, false, &part, &hp); if (r) return r; }
        { int r = vz_stream_ws(2, s, WIDE_TICKETS * sizeof(unsigned), true, &tick, &ht); if (r) return r; }
        if (!part || !tick || hp < WIDE_PART_BYTES || ht < WIDE_TICKETS * sizeof(unsigned)) {
            vz_set_error("wide gemm: no K-split scratch for this stream inside a capture (vz_wide_reserve before capturing)");
            return VZ_ERR_STATE;
        }
        p.part = (float*)part; p.ticket = (unsigned*)tick;
    }
    const bool sw = a.act == VZ_ACT_SWIGLU;
    const dim3 grid(rbs, p.P), block(NWV * 64);
    const size_t lds = (size_t)2 * CH * 2 * (a.M <= 32 ? 2 : 4) * 1024;
    if (a.W8t) {
        if (a.M <= 32) { if (sw) vz_launch_timed(wide_tiled_fp8_kernel<2, true>, grid, block, lds, s, p); else vz_launch_timed(wide_tiled_fp8_kernel<2, false>, grid, block, lds, s, p); }
        else { if (sw) vz_launch_timed(wide_tiled_fp8_kernel<4, true>, grid, block, lds, s, p); else vz_launch_timed(wide_tiled_fp8_kernel<4, false>, grid, block, lds, s, p); }
    } else if (a.M <= 32) {
        if (sw) vz_launch_timed(wide_tiled_kernel<2, true>, grid, block, lds, s, p);
        else vz_launch_timed(wide_tiled_kernel<2, false>, grid, block, lds, s, p);
    } else {
        if (sw) vz_launch_timed(wide_tiled_kernel<4, true>, grid, block, lds, s, p);
        else vz_launch_timed(wide_tiled_kernel<4, false>, grid, block, lds, s, p);
    }
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
