// One resident grid per decoded token (batch 1, Zephyr-7B geometry, bf16 weights): embedding row -> 32 x (RMSNorm + QKV, attention,
// O + residual, RMSNorm + gate|up SwiGLU, down + residual) -> final RMSNorm + lm_head logits, as PHASES of one launch instead of
// 161 launches (129 weight-streaming GEMVs + 32 attention kernels; hf:models/mistral/modeling_mistral.py:202-241 is the arithmetic).
//
// Why: every GEMV launch of the chain pays a ramp (first loads after the dependency, drain at the end: ~1.9 us each, 0.24 ms per
// token) and the attention chain (0.41 ms per token) leaves HBM idle.  Here 256 workgroups (one per CU, 12 waves) stay resident:
//   * waves 0..7 STREAM weights: 16-byte non-temporal loads straight to registers, two units (8..16 KiB each) in flight per wave at
//     all times - the loads of unit k+2 are issued right after unit k's dot products, ACROSS phase boundaries (a weight never depends
//     on the token), so the memory system keeps 128+ KiB per CU in flight while a phase edge is being crossed;
//   * wave 8 is the SYNC wave: it publishes the workgroup's results of a phase (LDS -> 4-byte write-through stores, drained, then ONE
//     agent-scope add on the workgroup's arrival counter shard), polls the 8 shards with L2-bypassing loads until every workgroup
//     has arrived, gathers the next phase's input vector with L2-bypassing 8-byte loads into LDS (fusing the RMSNorm) and releases
//     the streaming waves through a workgroup barrier.  It holds no weight loads, so its polls and gathers never queue behind a
//     prefetch (a wave's memory operations complete in issue order);
//   * waves 8..11 run the attention phase (attn_decode_body.h, the stand-alone kernel's body: same arithmetic, same bits) - they are
//     the only waves with registers to spare for it - while waves 0..7 keep the O-projection weights of the layer in flight.
// Hand-offs follow cdna guide section 6 G16 / the MI355X notes' table of measured forms: every handed-off byte is stored with sc1 by
// one wave that then drains (s_waitcnt vmcnt(0)); ONE lane of the workgroup adds to its shard after a workgroup barrier; consumers poll
// every shard with sc1 loads, then a workgroup barrier, then sc1 loads of the payload; one workgroup per CU; hipMalloc memory.
// Every spin is bounded: an expired wait raises the async error word and an abort word all other workgroups see, the launch ends,
// the logits of that token are garbage (the engine checks the word).  The counters are monotonic within a vz_llm_decode_steps call
// (target = (step * phases + phase + 1) * workgroups per shard) and zeroed by the host with the step counter.
//
// Arithmetic = the launch path's, bit for bit: a row's dot product runs over the same lane -> k assignment and chunk order as
// gemv_bf16_kernel, the RMSNorm statistics are summed in that kernel's order (4-wave form in front of QKV, 8-wave form elsewhere),
// epilogues round at the same points.  tests/test_persist_gpu.py holds both paths to equality.
#include <vector>

#include "attn_decode_body.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
using attn_dec::FusedParams;

constexpr int NSTREAM = 8;                 // streaming waves per workgroup
constexpr int NWAVES = 12;                 // + sync wave (8) + attention waves (8..11)
constexpr int TPB = NWAVES * 64;
constexpr int NWG = 256;                   // workgroups = CUs of an MI355X; the engine takes this path only on such a device
constexpr int NSW = NWG * NSTREAM;         // 2048 streaming waves
constexpr int HD = 4096, QKVN = 6144, IN = 14336;
constexpr int NQ = QKVN / NSW, NO = HD / NSW, NG = IN / NSW, ND = 2 * HD / NSW;     // units per wave and layer: 3, 2, 7, 4 (down: half rows)
static_assert(NQ == 3 && NO == 2 && NG == 7 && ND == 4, "unit schedule below is written for this geometry");
constexpr int DCH = IN / 512 / 2;          // chunks per half row of the down projection (14)
constexpr int NSHARD = 8;
constexpr int SPIN_CAP = 400000;           // bounded waits: ~0.2 s of polling, a legitimate wait is < 100 us
constexpr size_t LDS_REQUEST = 100 * 1024; // more than half a CU's 160 KiB: never two of these workgroups on one CU (each should own a CU's memory pipe)

struct TokLayer {
    const bf16_t *qkv_w, *o_w, *gu_w, *down_w;
    const float *in_norm, *post_norm;
    bf16_t *kc, *vc;
};

// the per-layer pointer table is read through the CONSTANT address space: scalar loads (lgkmcnt), never a vector load whose wait
// would drain the weight prefetch (vmcnt counts in issue order)
typedef const TokLayer* LayerTable;
__device__ __forceinline__ TokLayer load_layer(LayerTable t, int l) {
    static_assert(sizeof(TokLayer) == 8 * sizeof(unsigned long long), "eight pointers");
    const __attribute__((address_space(4))) unsigned long long* q = (const __attribute__((address_space(4))) unsigned long long*)(t + l);
    TokLayer L;
    L.qkv_w = (const bf16_t*)q[0]; L.o_w = (const bf16_t*)q[1]; L.gu_w = (const bf16_t*)q[2]; L.down_w = (const bf16_t*)q[3];
    L.in_norm = (const float*)q[4]; L.post_norm = (const float*)q[5]; L.kc = (bf16_t*)q[6]; L.vc = (bf16_t*)q[7];
    return L;
}
struct TokParams {
    const TokLayer* layers; int n_layers;
    const bf16_t* embed; const bf16_t* lm_head; const float* final_norm;
    const int* cur; const int* pos; const int* slot; const int* step;      // device-side decode state of row 0
    bf16_t *xa, *xb, *qkv, *att, *act;                                     // hand-off vectors (global, hipMalloc)
    float* logits;
    float* part; unsigned* ticket;                                         // attention partials / tickets
    const float *cosT, *sinT;
    unsigned* sync;                                                        // [NSHARD + 1][16] arrival counters, abort word
    unsigned long long* stamps;                                            // [n_layers][12] s_memrealtime of workgroup 0's sync wave (tools/persist_stamps.py)
    int* err;
    int vocab, max_ctx, nsplit, window;
    float scale, eps;
};

__device__ __forceinline__ unsigned long long ld8_sc1(const void* p) {
    return __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st4_sc1(void* p, unsigned v) {
    __hip_atomic_store((unsigned*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld4_sc1(const void* p) {
    return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gemv.hip's dot8: four sequential v_dot2c_f32_bf16 (same order: same bits)
__device__ __forceinline__ float dot8(const u32x4 w, const u32x4 x, float acc) {
    const unsigned w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w0), __builtin_bit_cast(bf16x2, x0), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w1), __builtin_bit_cast(bf16x2, x1), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w2), __builtin_bit_cast(bf16x2, x2), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, w3), __builtin_bit_cast(bf16x2, x3), acc, false);
    return acc;
}
// Pointers read from the per-layer table in memory carry no address space: hipcc would emit flat_load for them, whose completion
// order forces s_waitcnt vmcnt(0) lgkmcnt(0) at every use - no load could stay in flight across a unit.  A round trip through the
// global address space tells InferAddressSpaces what they are.
template <typename T>
__device__ __forceinline__ T* as_global(T* p) { return (T*)(__attribute__((address_space(1))) T*)p; }
__device__ __forceinline__ u32x4 ldw(const bf16_t* p) {
    return __builtin_nontemporal_load((const __attribute__((address_space(1))) u32x4*)p);
}

// LDS image of the workgroup
struct Lds {
    __attribute__((aligned(16))) bf16_t xn[IN];          // the phase's input vector as the dot products read it (normalised where the phase has a norm)
    __attribute__((aligned(16))) bf16_t xraw[HD];        // the residual stream the phase's epilogue adds (raw x of the layer / after the O projection)
    __attribute__((aligned(16))) unsigned short res[64]; // the streaming waves' outputs of the phase (<= 56), published by the sync wave
    int abort_flag;
    attn_dec::Shared attn;
};

enum { K_Q = 0, K_O = 1, K_G = 2, K_D = 3 };
// order pin: without it hipcc hoists the loads of several units above the dot products of the current one (400 registers, spills under the 168 this workgroup shape allows)
#define SB() __builtin_amdgcn_sched_barrier(0)

// ---- streaming side: the 16-byte loads of one unit ----
// K_Q / K_O: one row of K = 4096 (8 chunks of 512 k); K_G: the gate and the up row of one SwiGLU output (2 x 8); K_D: half a row of K = 14336 (14)
__device__ __forceinline__ void issue_row8(u32x4 (&wb)[16], const bf16_t* W, int row, int lane) {
    const bf16_t* wp = W + (size_t)row * HD + lane * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) wb[c] = ldw(wp + c * 512);
}
template <int KIND>
__device__ __forceinline__ void issue(u32x4 (&wb)[16], const TokLayer& L, int gw, int idx, int lane) {
    if (KIND == K_Q) issue_row8(wb, as_global(L.qkv_w), gw * NQ + idx, lane);
    if (KIND == K_O) issue_row8(wb, as_global(L.o_w), gw * NO + idx, lane);
    if (KIND == K_G) {
        const int j = gw * NG + idx;                                   // output j: gate row (j >> 4) * 32 + (j & 15), up row + 16 ([16 gate | 16 up] interleave)
        const bf16_t* wg = as_global(L.gu_w) + (size_t)((j >> 4) * 32 + (j & 15)) * HD + lane * 8;
#pragma unroll
        for (int c = 0; c < 8; ++c) wb[c] = ldw(wg + c * 512);
#pragma unroll
        for (int c = 0; c < 8; ++c) wb[8 + c] = ldw(wg + (size_t)16 * HD + c * 512);
    }
    if (KIND == K_D) {
        const bf16_t* wp = as_global(L.down_w) + (size_t)(gw * NO + (idx >> 1)) * IN + (size_t)(idx & 1) * DCH * 512 + lane * 8;
#pragma unroll
        for (int c = 0; c < DCH; ++c) wb[c] = ldw(wp + c * 512);
    }
}

// ---- streaming side: dot products + epilogue of one unit; results go to LDS (the sync wave publishes them) ----
__device__ __forceinline__ float dot_chunks8(const u32x4 (&wb)[16], int off, const bf16_t* xn, int lane, float acc) {
#pragma unroll
    for (int c = 0; c < 8; ++c) acc = dot8(wb[off + c], *(const u32x4*)(xn + c * 512 + lane * 8), acc);
    return acc;
}
template <int KIND>
__device__ __forceinline__ void compute(const u32x4 (&wb)[16], Lds& sm, int wave, int gw, int idx, int lane, float& carry) {
    if (KIND == K_Q) {
        const float r = wave_sum_lane63(dot_chunks8(wb, 0, sm.xn, lane, 0.f));
        if (lane == 63) sm.res[wave * NQ + idx] = f32_to_bf16(r);
    }
    if (KIND == K_O) {
        const float r = wave_sum_lane63(dot_chunks8(wb, 0, sm.xn, lane, 0.f));
        if (lane == 63) sm.res[wave * NO + idx] = f32_to_bf16(r + bf16_to_f32(sm.xraw[gw * NO + idx]));
    }
    if (KIND == K_G) {
        const float g = wave_sum_lane63(dot_chunks8(wb, 0, sm.xn, lane, 0.f));
        const float u = wave_sum_lane63(dot_chunks8(wb, 8, sm.xn, lane, 0.f));
        if (lane == 63) sm.res[wave * NG + idx] = f32_to_bf16(act_silu(g) * u);
    }
    if (KIND == K_D) {
        float acc = (idx & 1) ? carry : 0.f;
        const bf16_t* xh = sm.xn + (idx & 1) * DCH * 512;
#pragma unroll
        for (int c = 0; c < DCH; ++c) acc = dot8(wb[c], *(const u32x4*)(xh + c * 512 + lane * 8), acc);
        carry = acc;
        if (idx & 1) {
            const float r = wave_sum_lane63(acc);
            if (lane == 63) sm.res[wave * NO + (idx >> 1)] = f32_to_bf16(r + bf16_to_f32(sm.xraw[gw * NO + (idx >> 1)]));
        }
    }
}

// ---- sync wave ----
__device__ __forceinline__ void stamp(const TokParams& p, int wg, int lane, int l, int k) {
    if (wg == 0 && lane == 0) p.stamps[l * 12 + k] = __builtin_amdgcn_s_memrealtime();
}
// arrive: this workgroup has finished phase `seq` (all its stores are drained: the caller sits behind the workgroup barrier)
__device__ __forceinline__ void arrive(const TokParams& p, int wg, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(p.sync + (wg & (NSHARD - 1)) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until every workgroup has arrived `target / (NWG / NSHARD)` times; bounded; returns false after an abort
__device__ __forceinline__ bool wait_all(const TokParams& p, Lds& sm, unsigned target, int lane) {
    if (sm.abort_flag) return false;
    const unsigned* w = p.sync + (lane < NSHARD ? lane : NSHARD) * 16;      // lanes 0..7: a shard each; the others: the abort word
    for (int it = 0; it < SPIN_CAP; ++it) {
        const unsigned v = ld4_sc1(w);
        const bool ok = lane < NSHARD ? (int)(v - target) >= 0 : true;      // (wrap-safe: the counters only grow)
        const bool ab = lane >= NSHARD && v != 0u;
        if (__builtin_amdgcn_ballot_w64(ab) != 0ull) {                      // another workgroup gave up (or the host aborted): so do we
            if (lane == 0) atomicExch(p.err, VZ_ASYNC_PERSIST);
            sm.abort_flag = 1;
            return false;
        }
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    if (lane == 0) {
        __hip_atomic_store(p.sync + NSHARD * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(p.err, VZ_ASYNC_PERSIST);
    }
    sm.abort_flag = 1;
    return false;
}
// publish n bf16 results of this workgroup (LDS res[0..n)) at dst[wg * n ..]: 4-byte write-through stores, drained
__device__ __forceinline__ void publish(bf16_t* dst, const Lds& sm, int wg, int n, int lane) {
    if (lane < n / 2) st4_sc1(dst + (size_t)wg * n + lane * 2, (unsigned)sm.res[lane * 2] | ((unsigned)sm.res[lane * 2 + 1] << 16));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// gather n bf16 (n % 256 == 0) from a hand-off vector into LDS: 8-byte L2-bypassing loads, all in flight before the first use
template <int N, bool SC1>
__device__ __forceinline__ void gather_raw(const bf16_t* src, bf16_t* dst, int lane) {
    constexpr int NI = N / 256;
    constexpr int PIECE = NI <= 28 ? NI : 28;           // <= 56 registers of payload in flight (the kernel runs 12 waves per CU: 168 per lane)
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += PIECE) {
        unsigned long long v[PIECE];
#pragma unroll
        for (int i = 0; i < PIECE; ++i)
            if (i0 + i < NI) v[i] = SC1 ? ld8_sc1(src + (i0 + i) * 256 + lane * 4) : *(const unsigned long long*)(src + (i0 + i) * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < PIECE; ++i)
            if (i0 + i < NI) *(unsigned long long*)(dst + (i0 + i) * 256 + lane * 4) = v[i];
    }
}
// RMSNorm of the 4096-wide raw vector in LDS (xraw) into xn, statistics summed in gemv_bf16_kernel's order: NW-wave workgroup, thread t
// squares its 8-element groups k = 8 t + 8 * 64 * NW * j in order, wave_sum per wave, the waves' sums added in wave order
template <int NW>
__device__ __forceinline__ void rmsnorm_lds(Lds& sm, const float* norm_w, float eps, int lane) {
    float tot = 0.f;
#pragma unroll
    for (int vw = 0; vw < NW; ++vw) {
        float ss = 0.f;
        for (int k = (vw * 64 + lane) * 8; k < HD; k += NW * 64 * 8) {
            const u16x8 v = *(const u16x8*)(sm.xraw + k);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(v[j]); ss += f * f; }
        }
        tot += wave_sum(ss);
    }
    const float rstd = rsqrtf(tot / (float)HD + eps);
    for (int k = lane * 8; k < HD; k += 64 * 8) {
        const u16x8 v = *(const u16x8*)(sm.xraw + k);
        const f32x4 w0 = *(const f32x4*)(norm_w + k), w1 = *(const f32x4*)(norm_w + k + 4);
        u16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float wj = j < 4 ? w0[j] : w1[j - 4];
            o[j] = f32_to_bf16(wj * (bf16_to_f32(v[j]) * rstd));
        }
        *(u16x8*)(sm.xn + k) = o;
    }
}

// ---- the two roles.  They are separate functions on purpose: the streaming waves keep 128 registers of weights live across every
//      barrier, the attention body needs ~150 of its own; in one control-flow graph the allocator would have to hold both at once. ----
// Barrier ledger per layer (every wave executes exactly these, in this order):
//   S1 inputs of QKV ready | S2 QKV results in LDS | S3 q|k|v row gathered | [attention body's barriers] | S4 attention stores drained |
//   S5 attention output gathered | S6 O results in LDS | S7 inputs of gate|up ready | S8 gate|up results in LDS | S9 inputs of down ready |
//   S10 down results in LDS;   after the last layer: S11 inputs of the lm_head ready.
__device__ __forceinline__ void run_streamer(const TokParams& p, Lds& sm, const FusedParams& ap0, int wg, int wave, int lane, bool a_on, int a_split) {
    const int gw = wg * NSTREAM + wave;
    const int L = p.n_layers;
    // lm_head rows of this wave: workgroup w owns rows [w V / NWG, (w + 1) V / NWG), split over its 8 streaming waves the same way
    const int lm_w0 = (int)((long)wg * p.vocab / NWG), lm_w1 = (int)((long)(wg + 1) * p.vocab / NWG);
    const int lm_r0 = lm_w0 + (int)((long)wave * (lm_w1 - lm_w0) / NSTREAM), lm_r1 = lm_w0 + (int)((long)(wave + 1) * (lm_w1 - lm_w0) / NSTREAM);
    const int lm_last = min(max(lm_r0, lm_r1 - 1), p.vocab - 1);
    const int slot0 = __builtin_amdgcn_readfirstlane(p.slot[0]);     // (before any weight load: a later vector load would wait for the whole prefetch)
    u32x4 wb0[16], wb1[16];
    float carry = 0.f;
    const LayerTable table = (LayerTable)p.layers;
    {
        const TokLayer L0 = load_layer(table, 0);
        issue<K_Q>(wb0, L0, gw, 0, lane);               // the first two units of layer 0 go in flight before anything else
        issue<K_Q>(wb1, L0, gw, 1, lane);
    }
    __syncthreads();                                    // S0: abort flag initialised
    // Schedule of a phase with n units per wave (unit 0 in wb0, unit 1 in wb1 when it starts): compute(k), issue(k + 2) for k < n - 2,
    // then the last two computes back to back, the "results in LDS" barrier, and only THEN the first two units of the next phase:
    // issuing a unit blocks the wave until the CU's memory pipe has taken its 8..16 KiB (measured: ~2 us per 16-KiB unit with all eight
    // waves issuing - the pipe accepts at about the CU's HBM share), and nobody may wait for that in front of the barrier that lets
    // the sync wave publish.  The two units issued behind it are what keeps HBM busy while the phase edge is crossed.
    for (int l = 0; l < L; ++l) {
        const TokLayer Ly = load_layer(table, l);
        const bf16_t* next_qkv = load_layer(table, l + 1 < L ? l + 1 : l).qkv_w;
        __syncthreads();                                // S1
        compute<K_Q>(wb0, sm, wave, gw, 0, lane, carry); SB(); issue<K_Q>(wb0, Ly, gw, 2, lane); SB();
        compute<K_Q>(wb1, sm, wave, gw, 1, lane, carry); SB();
        compute<K_Q>(wb0, sm, wave, gw, 2, lane, carry); SB();
        __syncthreads();                                // S2
        issue<K_O>(wb0, Ly, gw, 0, lane); SB(); issue<K_O>(wb1, Ly, gw, 1, lane); SB();
        __syncthreads();                                // S3
        if (a_on) attn_dec::shadow(ap0, a_split, slot0, sm.attn);     // the attention body's barriers; wb0 / wb1 hold the O projection's rows meanwhile
        __syncthreads();                                // S4
        __syncthreads();                                // S5
        compute<K_O>(wb0, sm, wave, gw, 0, lane, carry); SB();
        compute<K_O>(wb1, sm, wave, gw, 1, lane, carry); SB();
        __syncthreads();                                // S6
        issue<K_G>(wb0, Ly, gw, 0, lane); SB(); issue<K_G>(wb1, Ly, gw, 1, lane); SB();
        __syncthreads();                                // S7
        compute<K_G>(wb0, sm, wave, gw, 0, lane, carry); SB(); issue<K_G>(wb0, Ly, gw, 2, lane); SB();
        compute<K_G>(wb1, sm, wave, gw, 1, lane, carry); SB(); issue<K_G>(wb1, Ly, gw, 3, lane); SB();
        compute<K_G>(wb0, sm, wave, gw, 2, lane, carry); SB(); issue<K_G>(wb0, Ly, gw, 4, lane); SB();
        compute<K_G>(wb1, sm, wave, gw, 3, lane, carry); SB(); issue<K_G>(wb1, Ly, gw, 5, lane); SB();
        compute<K_G>(wb0, sm, wave, gw, 4, lane, carry); SB(); issue<K_G>(wb0, Ly, gw, 6, lane); SB();
        compute<K_G>(wb1, sm, wave, gw, 5, lane, carry); SB();
        compute<K_G>(wb0, sm, wave, gw, 6, lane, carry); SB();
        __syncthreads();                                // S8
        issue<K_D>(wb0, Ly, gw, 0, lane); SB(); issue<K_D>(wb1, Ly, gw, 1, lane); SB();
        __syncthreads();                                // S9
        compute<K_D>(wb0, sm, wave, gw, 0, lane, carry); SB(); issue<K_D>(wb0, Ly, gw, 2, lane); SB();
        compute<K_D>(wb1, sm, wave, gw, 1, lane, carry); SB(); issue<K_D>(wb1, Ly, gw, 3, lane); SB();
        compute<K_D>(wb0, sm, wave, gw, 2, lane, carry); SB();
        compute<K_D>(wb1, sm, wave, gw, 3, lane, carry); SB();
        __syncthreads();                                // S10
        // the next two units belong to the next layer's QKV projection - or to the lm_head behind the last layer (same shape: rows of K = 4096)
        const bool more = l + 1 < L;
        const bf16_t* nw = more ? as_global(next_qkv) : p.lm_head;
        const int nr0 = more ? gw * NQ : min(lm_r0, lm_last), nr1 = more ? gw * NQ + 1 : min(lm_r0 + 1, lm_last);
        issue_row8(wb0, nw, nr0, lane); SB(); issue_row8(wb1, nw, nr1, lane); SB();
    }
    __syncthreads();                                    // S11
    // lm_head: rows lm_r0 .. lm_r1 - 1, two in flight (wb0 / wb1 hold the first two); rows past the wave's range are clamped and not stored
    for (int r = lm_r0; r < lm_r1; r += 2) {
        {
            const float v = wave_sum_lane63(dot_chunks8(wb0, 0, sm.xn, lane, 0.f));
            if (lane == 63) p.logits[r] = v;
            SB(); issue_row8(wb0, p.lm_head, min(r + 2, lm_last), lane); SB();
        }
        {
            const float v = wave_sum_lane63(dot_chunks8(wb1, 0, sm.xn, lane, 0.f));
            if (lane == 63 && r + 1 < lm_r1) p.logits[r + 1] = v;
            SB(); issue_row8(wb1, p.lm_head, min(r + 3, lm_last), lane); SB();
        }
    }
}

__device__ __forceinline__ void run_sync_attn(const TokParams& p, Lds& sm, FusedParams ap, int wg, int wave, const int lane_in, const int a_tid_in, bool a_on, int a_split, int a_hk) {
    const int lane = lane_in, a_tid = a_tid_in;
    const bool syncer = wave == NSTREAM;
    const int L = p.n_layers;
    const unsigned per_shard = NWG / NSHARD;
    const unsigned base = (unsigned)p.step[0] * (unsigned)(5 * L);       // phases completed by earlier tokens of this vz_llm_decode_steps call
    __syncthreads();                                    // S0
    const int lane0 = lane, a_tid0 = a_tid;
    for (int l = 0; l < L; ++l) {
        // per-lane values are made opaque once per layer: otherwise hipcc hoists every address the roles below derive from them out
        // of the layer loop, keeps ~120 of them live across it and spills them (the budget is 168 registers per lane)
        int lane = lane0, a_tid = a_tid0;
        asm volatile("" : "+v"(lane), "+v"(a_tid));
        const TokLayer Ly = load_layer(p.layers, l);
        const unsigned q0 = base + (unsigned)l * 5;
        if (syncer) {
            stamp(p, wg, lane, l, 0);
            if (l == 0) gather_raw<HD, false>(p.embed + (size_t)p.cur[0] * HD, sm.xraw, lane);
            else { (void)wait_all(p, sm, q0 * per_shard, lane); stamp(p, wg, lane, l, 1); gather_raw<HD, true>(p.xa, sm.xraw, lane); }
            rmsnorm_lds<4>(sm, as_global(Ly.in_norm), p.eps, lane);
            stamp(p, wg, lane, l, 2);
        }
        __syncthreads();                                // S1
        __syncthreads();                                // S2
        if (syncer) {
            stamp(p, wg, lane, l, 3);
            publish(p.qkv, sm, wg, NSTREAM * NQ, lane);
            arrive(p, wg, lane);
            (void)wait_all(p, sm, (q0 + 1) * per_shard, lane);
            gather_raw<QKVN, true>(p.qkv, sm.xn, lane);
            stamp(p, wg, lane, l, 4);
        }
        __syncthreads();                                // S3
        if (a_on) {
            ap.kc = as_global(Ly.kc); ap.vc = as_global(Ly.vc);
            (void)attn_dec::body<true, true>(ap, sm.xn, a_split, a_hk, 0, a_tid, true, sm.attn);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // partial / merged stores of the attention waves are drained
        }
        __syncthreads();                                // S4
        if (syncer) {
            stamp(p, wg, lane, l, 5);
            arrive(p, wg, lane);
            (void)wait_all(p, sm, (q0 + 2) * per_shard, lane);
            gather_raw<HD, true>(p.att, sm.xn, lane);
            stamp(p, wg, lane, l, 6);
        }
        __syncthreads();                                // S5
        __syncthreads();                                // S6
        if (syncer) {
            stamp(p, wg, lane, l, 7);
            publish(p.xb, sm, wg, NSTREAM * NO, lane);
            arrive(p, wg, lane);
            (void)wait_all(p, sm, (q0 + 3) * per_shard, lane);
            gather_raw<HD, true>(p.xb, sm.xraw, lane);
            rmsnorm_lds<8>(sm, as_global(Ly.post_norm), p.eps, lane);
            stamp(p, wg, lane, l, 8);
        }
        __syncthreads();                                // S7
        __syncthreads();                                // S8
        if (syncer) {
            stamp(p, wg, lane, l, 9);
            publish(p.act, sm, wg, NSTREAM * NG, lane);
            arrive(p, wg, lane);
            (void)wait_all(p, sm, (q0 + 4) * per_shard, lane);
            gather_raw<IN, true>(p.act, sm.xn, lane);
            stamp(p, wg, lane, l, 10);
        }
        __syncthreads();                                // S9
        __syncthreads();                                // S10
        if (syncer) {
            stamp(p, wg, lane, l, 11);
            publish(p.xa, sm, wg, NSTREAM * NO, lane);
            arrive(p, wg, lane);
        }
    }
    if (syncer) {
        (void)wait_all(p, sm, (base + (unsigned)L * 5) * per_shard, lane0);
        gather_raw<HD, true>(p.xa, sm.xraw, lane0);
        rmsnorm_lds<8>(sm, p.final_norm, p.eps, lane0);
    }
    __syncthreads();                                    // S11
}

__global__ __launch_bounds__(TPB) void decode_token_kernel(TokParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Lds& sm = *reinterpret_cast<Lds*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    if (tid == 0) sm.abort_flag = 0;
    // attention geometry of this workgroup: (context split, KV head) as the stand-alone kernel's grid (nsplit, Hkv, 1)
    FusedParams ap;
    ap.qkv = nullptr; ap.kc = nullptr; ap.vc = nullptr; ap.o = p.att; ap.part = p.part; ap.ticket = p.ticket; ap.cosT = p.cosT; ap.sinT = p.sinT;
    ap.pos = p.pos; ap.slot = p.slot;
    ap.B = 1; ap.Hq = 32; ap.Hkv = 8; ap.max_ctx = p.max_ctx; ap.nsplit = p.nsplit; ap.window = p.window; ap.scale = p.scale;
    const int a_split = wg % p.nsplit, a_hk = wg / p.nsplit;
    const bool a_on = a_hk < 8;
    if (wave < NSTREAM) run_streamer(p, sm, ap, wg, wave, lane, a_on, a_split);
    else run_sync_attn(p, sm, ap, wg, wave, lane, tid - NSTREAM * 64, a_on, a_split, a_hk);
}

}  // namespace

// ---- host side ----
struct VzTokState {
    TokLayer* d_layers = nullptr; int n_layers = 0;
    bf16_t* d_vec = nullptr;             // xa | xb | qkv | att | act
    unsigned* d_sync = nullptr;
    unsigned long long* d_stamps = nullptr;
};

bool vz_decode_persist_supported() {
    static int ok = -1;
    if (ok >= 0) return ok == 1;
    ok = 0;
    int dev = 0, cus = 0, blocks = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus != NWG) return false;
    if (hipFuncSetAttribute((const void*)decode_token_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_REQUEST) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, (const void*)decode_token_kernel, TPB, LDS_REQUEST) != hipSuccess || blocks < 1) { (void)hipGetLastError(); return false; }
    ok = 1;
    return true;
}

int vz_decode_persist_create(const VzTokLayerHost* layers, int n_layers, VzTokState** out) {
    VZ_CHECK_ARG(layers && n_layers >= 1 && out, "decode_persist_create: bad argument");
    VzTokState* st = new VzTokState();
    std::vector<TokLayer> h(n_layers);
    for (int i = 0; i < n_layers; ++i) {
        h[i].qkv_w = (const bf16_t*)layers[i].qkv_w; h[i].o_w = (const bf16_t*)layers[i].o_w; h[i].gu_w = (const bf16_t*)layers[i].gu_w; h[i].down_w = (const bf16_t*)layers[i].down_w;
        h[i].in_norm = layers[i].in_norm; h[i].post_norm = layers[i].post_norm; h[i].kc = (bf16_t*)layers[i].kc; h[i].vc = (bf16_t*)layers[i].vc;
    }
    hipError_t er = hipMalloc((void**)&st->d_layers, n_layers * sizeof(TokLayer));
    if (er == hipSuccess) er = hipMemcpy(st->d_layers, h.data(), n_layers * sizeof(TokLayer), hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMalloc((void**)&st->d_vec, (size_t)(HD + HD + QKVN + HD + IN) * sizeof(bf16_t));
    if (er == hipSuccess) er = hipMalloc((void**)&st->d_sync, (NSHARD + 1) * 16 * sizeof(unsigned));
    if (er == hipSuccess) er = hipMemset(st->d_sync, 0, (NSHARD + 1) * 16 * sizeof(unsigned));
    if (er == hipSuccess) er = hipMalloc((void**)&st->d_stamps, (size_t)n_layers * 12 * sizeof(unsigned long long));
    if (er == hipSuccess) er = hipMemset(st->d_stamps, 0, (size_t)n_layers * 12 * sizeof(unsigned long long));
    if (er != hipSuccess) { vz_set_error("decode_persist_create: %s", hipGetErrorString(er)); vz_decode_persist_destroy(st); return VZ_ERR_HIP; }
    st->n_layers = n_layers;
    *out = st;
    return VZ_OK;
}

void vz_decode_persist_destroy(VzTokState* st) {
    if (!st) return;
    if (st->d_layers) (void)hipFree(st->d_layers);
    if (st->d_vec) (void)hipFree(st->d_vec);
    if (st->d_sync) (void)hipFree(st->d_sync);
    if (st->d_stamps) (void)hipFree(st->d_stamps);
    delete st;
}

// zero the arrival counters / abort word: together with the step counter, at the start of every vz_llm_decode_steps call
int vz_decode_persist_reset(VzTokState* st, hipStream_t s) {
    VZ_CHECK_HIP(hipMemsetAsync(st->d_sync, 0, (NSHARD + 1) * 16 * sizeof(unsigned), s));
    return VZ_OK;
}

// TEST HOOK: preset shard 0's counter so that a wait can never be satisfied in order / is satisfied early
int vz_decode_persist_poke(VzTokState* st, int word, unsigned value, hipStream_t s) {
    VZ_CHECK_ARG(st && word >= 0 && word <= NSHARD, "decode_persist_poke: bad word");
    VZ_CHECK_HIP(hipMemcpyAsync(st->d_sync + word * 16, &value, sizeof(unsigned), hipMemcpyHostToDevice, s));
    VZ_CHECK_HIP(hipStreamSynchronize(s));
    return VZ_OK;
}

// the phase stamps of the LAST launched token (workgroup 0's sync wave; 12 per layer, 100 MHz ticks): blocking copy
int vz_decode_persist_stamps(VzTokState* st, unsigned long long* host, int n_layers) {
    VZ_CHECK_ARG(st && host && n_layers == st->n_layers, "decode_persist_stamps: bad argument");
    VZ_CHECK_HIP(hipDeviceSynchronize());
    VZ_CHECK_HIP(hipMemcpy(host, st->d_stamps, (size_t)n_layers * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return VZ_OK;
}

int vz_launch_decode_token(VzTokState* st, const VzTokArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(st && a.vocab >= NWG && a.nsplit >= 1 && a.nsplit <= 32, "decode_token: bad argument (nsplit %d)", a.nsplit);
    TokParams p;
    p.layers = st->d_layers; p.n_layers = st->n_layers;
    p.embed = (const bf16_t*)a.embed; p.lm_head = (const bf16_t*)a.lm_head; p.final_norm = a.final_norm;
    p.cur = a.cur; p.pos = a.pos; p.slot = a.slot; p.step = a.step;
    p.xa = st->d_vec; p.xb = p.xa + HD; p.qkv = p.xb + HD; p.att = p.qkv + QKVN; p.act = p.att + HD;
    p.logits = a.logits; p.part = a.part; p.ticket = a.ticket; p.cosT = a.cosT; p.sinT = a.sinT;
    p.sync = st->d_sync; p.stamps = st->d_stamps; p.err = a.err;
    p.vocab = a.vocab; p.max_ctx = a.max_ctx; p.nsplit = a.nsplit; p.window = a.window; p.scale = a.scale; p.eps = a.eps;
    static_assert(sizeof(Lds) <= LDS_REQUEST, "LDS image");
    vz_launch_timed(decode_token_kernel, dim3(NWG), dim3(TPB), LDS_REQUEST, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
