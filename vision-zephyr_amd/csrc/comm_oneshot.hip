// One-shot all-reduce for the tensor-parallel DECODE step (SURVEY.md section 5 last row, section 8e; north_star: "Zephyr tensor-parallel
// shards ... RCCL all-reduce/all-gather over xGMI").  A TP decode step carries 64 all-reduces of [B, 4096] bf16 (8 KB at B = 1): pure
// latency.  RCCL's generic ring / tree costs ~15-25 us per call at N = 8; xGMI is a full mesh of point-to-point
// links, so each rank can hand its partial vector to every peer DIRECTLY:
//
//   every rank r, for every peer q (itself included): store its vector into q's receive area slot[seq & 1][r] as 8-byte GRANULES
//   {two bf16 of data, 32-bit tag = seq} (one naturally aligned 8-byte store each: data and tag can never be seen apart - the
//   MI355X notes' "R2 granule"), system scope so the bytes leave through xGMI; then it sweeps its OWN area until the tag of every
//   granule of every rank equals seq, and adds the N vectors in RANK ORDER in fp32 (every rank computes the same bits; the residual
//   rides in rank 0's partial as before).  No flag, no fence, no second hop: one store-and-poll round per all-reduce.
//   Two slots by seq parity: a rank can run at most one all-reduce ahead of a peer (finishing s + 1 needs every peer's s + 1 vector,
//   which a peer only sends after it has read message s), so message s + 2 never overwrites an unread s.
//
// The sweep is bounded (an absent peer raises the async error word, the output is poisoned with NaN).  The sequence number lives in
// device memory and advances inside the kernel, so the launch replays from a captured hipGraph.
// Multi-process wiring (peer areas opened through hipIpc handles exchanged over torch.distributed) is vz_hip/tp.py's; this file
// only needs N device pointers.  No multi-GPU box has been available to the build: the protocol and arithmetic are tested in ONE
// process with N areas on one GPU and N concurrent launches standing in for N ranks (tests/test_oneshot_gpu.py); its time over
// xGMI is unmeasured.
#include "vz_common.h"

namespace {

constexpr int OS_MAX_RANKS = 8;
constexpr int OS_SPIN_CAP = 300000;      // ~0.3 s; a message over xGMI takes microseconds

struct OsParams {
    unsigned long long* area[OS_MAX_RANKS];   // receive area of every rank (peer-mapped); own = area[rank]
    const bf16_t* in[OS_MAX_RANKS];           // the partial [n] of rank (rank0 + blockIdx.y)
    bf16_t* out[OS_MAX_RANKS];                // its reduced vector [n] (may alias in)
    unsigned* seq[OS_MAX_RANKS];              // its device words {sequence number of the NEXT all-reduce (advanced by this launch), ticket}
    int* err;
    int rank0, n_here, n_ranks, n;            // n = bf16 elements, multiple of 2.  Production: n_here = 1, rank0 = the rank.  One-process test:
                                              // n_here = n_ranks, rank0 = 0 - the ranks are interleaved slices of ONE grid (block b = rank b % n_here,
                                              // granule block b / n_here: whatever part of the grid is resident holds every rank of its granule blocks;
                                              // N launches on N streams of one GPU may share a hardware queue and then wait for each other for ever)
    int cap;                                  // granules per (slot, rank) of an area
};

__device__ __forceinline__ void st8_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld8_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// grid: ceil(n / 2 / 256) workgroups of 256 threads, one granule (two bf16) per thread
__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(OsParams p) {
    const int ri = blockIdx.x % p.n_here, gb = blockIdx.x / p.n_here;
    const int g = gb * 256 + threadIdx.x;                    // granule index
    const int ng = p.n >> 1;
    const int rank = p.rank0 + ri;
    unsigned* seqw = p.seq[ri];
    const bf16_t* in = p.in[ri];
    bf16_t* out = p.out[ri];
    const unsigned seq = __hip_atomic_load(seqw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // every thread reads it before anyone advances it (see below)
    const size_t slot = (size_t)(seq & 1u) * p.n_ranks * p.cap;
    if (g < ng) {
        const unsigned data = *(const unsigned*)(in + 2 * g);
        const unsigned long long gran = (unsigned long long)data | ((unsigned long long)seq << 32);
        for (int q = 0; q < p.n_ranks; ++q) st8_sys(p.area[q] + slot + (size_t)rank * p.cap + g, gran);
    }
    float a0 = 0.f, a1 = 0.f;
    bool ok = true;
    if (g < ng) {
        const unsigned long long* mine = p.area[rank] + slot + g;
        for (int r = 0; r < p.n_ranks && ok; ++r) {           // rank order: the same fp32 sum on every rank
            unsigned long long v = 0;
            int it = 0;
            for (; it < OS_SPIN_CAP; ++it) {
                v = ld8_sys(mine + (size_t)r * p.cap);
                if ((unsigned)(v >> 32) == seq) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (it == OS_SPIN_CAP) { ok = false; break; }
            a0 += bf16_to_f32((unsigned short)(v & 0xffffu));
            a1 += bf16_to_f32((unsigned short)((v >> 16) & 0xffffu));
        }
        if (!ok) { atomicExch(p.err, VZ_ASYNC_ONESHOT); a0 = a1 = __uint_as_float(0x7fc00000u); }
        *(unsigned*)(out + 2 * g) = pack_bf16x2(a0, a1);
    }
    // advance the sequence number once per launch: the last workgroup to get here does it (every thread of every workgroup has read
    // `seq` above, before its own workgroup arrived) - a ticket in the word next to it
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(seqw + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x / p.n_here - 1) {
            __hip_atomic_store(seqw + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(seqw, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// bytes of one rank's receive area for vectors of up to max_elems bf16: [2 slots][n_ranks][max_elems / 2] granules
size_t vz_oneshot_area_bytes(int n_ranks, int max_elems) { return (size_t)2 * n_ranks * (max_elems / 2) * sizeof(unsigned long long); }

static int os_launch(void* const* areas, int rank0, int n_here, int n_ranks, int max_elems, const bf16_t* const* in, bf16_t* const* out, int n,
                     unsigned* const* seq, int* err, hipStream_t s) {
    VZ_CHECK_ARG(areas && in && out && seq && err && n_ranks >= 1 && n_ranks <= OS_MAX_RANKS && rank0 >= 0 && rank0 + n_here <= n_ranks && n_here >= 1,
                 "allreduce_oneshot: bad argument (ranks %d, rank %d)", n_ranks, rank0);
    VZ_CHECK_ARG(n >= 2 && (n & 1) == 0 && n <= max_elems && (max_elems & 1) == 0, "allreduce_oneshot: n = %d must be even and <= the area's %d elements", n, max_elems);
    OsParams p;
    for (int q = 0; q < OS_MAX_RANKS; ++q) { p.area[q] = q < n_ranks ? (unsigned long long*)areas[q] : nullptr; p.in[q] = nullptr; p.out[q] = nullptr; p.seq[q] = nullptr; }
    for (int q = 0; q < n_ranks; ++q) VZ_CHECK_ARG(areas[q] && ((uintptr_t)areas[q] & 7) == 0, "allreduce_oneshot: area %d missing / misaligned", q);
    for (int i = 0; i < n_here; ++i) {
        VZ_CHECK_ARG(in[i] && out[i] && seq[i] && ((uintptr_t)in[i] & 3) == 0 && ((uintptr_t)out[i] & 3) == 0, "allreduce_oneshot: vectors must be 4-byte aligned");
        p.in[i] = in[i]; p.out[i] = out[i]; p.seq[i] = seq[i];
    }
    p.err = err; p.rank0 = rank0; p.n_here = n_here; p.n_ranks = n_ranks; p.n = n; p.cap = max_elems / 2;
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(((n / 2 + 255) / 256) * n_here), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}

int vz_launch_allreduce_oneshot(void* const* areas, int rank, int n_ranks, int max_elems, const bf16_t* in, bf16_t* out, int n, unsigned* seq,
                                int* err, hipStream_t s) {
    return os_launch(areas, rank, 1, n_ranks, max_elems, &in, &out, n, &seq, err, s);
}

// ONE-PROCESS TEST FORM: all n_ranks ranks as slices (blockIdx.y) of one grid - co-resident by construction
int vz_launch_allreduce_oneshot_all(void* const* areas, int n_ranks, int max_elems, const bf16_t* const* in, bf16_t* const* out, int n,
                                    unsigned* const* seq, int* err, hipStream_t s) {
    return os_launch(areas, 0, n_ranks, n_ranks, max_elems, in, out, n, seq, err, s);
}
