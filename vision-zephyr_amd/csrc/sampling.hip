// Device-side sampling tail of a decode step (SURVEY.md section 8a', last Zephyr row; a13):
//   hf:generation/utils.py `_sample` with do_sample=True = logits[:, -1].float() -> TemperatureLogitsWarper (x / T) ->
//   TopKLogitsWarper (keep x >= k-th largest; HF's GenerationConfig default top_k = 50 applies on the reference's CLI path,
//   ref:vis_zephyr/serve/cli.py:171-182 passes only do_sample + temperature) -> TopPLogitsWarper (sort ascending, drop the
//   prefix whose cumulative softmax mass is <= 1 - top_p, keep >= 1) -> softmax -> torch.multinomial(probs, 1).
//
// One 1024-thread workgroup per row; the row (128 KB of fp32 logits) stays in L2 across the passes:
//   1. max of x = logit / T                                   (division, as the warper divides)
//   2. top-k: exact k-th largest by a 4-pass radix select on the order-preserving key of x (256-bin LDS histograms)
//   3. top-p: Z = sum of w_i = floor(exp(x_i - max) * 2^40) over the kept tokens (integer mass: order-independent, so a
//      replayed graph, an eager step and a second run give the same bits), then a 4-pass radix select ASCENDING on the
//      same key with the histogram weighted by w: smallest key t with mass{key <= t} > (1 - top_p) * Z; keep key >= t
//   4. draw: token = argmax_i (x_i + G_i) over the kept tokens, G_i = -log(-log(u_i)) (Gumbel race: P[argmax = i] =
//      softmax(x)_i, the distribution torch.multinomial draws from), u_i = Philox4x32-10(key = seed, counter =
//      (i, row, token counter, 0x565a)) word 0 -> ((w >> 9) + 0.5) * 2^-23; ties -> smallest index.
// torch.multinomial's own bit stream is a property of torch's generator, not of the reference; what is pinned is the
// distribution (tests/test_sampling_gpu.py: chi-square against the warped softmax) and the exact draw against the numpy
// restatement oracle/sampling_oracle.py (same Philox stream).
// The same kernel is the step's tail inside the per-token hipGraph: it publishes the token for the next step, appends it
// to the output ids and advances the per-row position / cache slot / length kept on the device (as argmax_kernel does).
#include "vz_common.h"

namespace {

__device__ __forceinline__ unsigned fkey(float x) {          // ascending-order-preserving key of a float
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned philox_word0(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

struct SampleShared {
    unsigned hist[256];
    unsigned long long mass[256];
    float redf[16];
    int redi[16];
    unsigned long long redm[16];
    unsigned sel;                 // selected radix bin of the current pass
    unsigned long long carry;     // count / mass still to be found inside the selected bin
};

__device__ __forceinline__ float block_max(float v, SampleShared& sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh.redf[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh.redf[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) r = fmaxf(r, sh.redf[w]);
    __syncthreads();
    return r;
}

// REG (cols <= 32768): the tempered row is staged ONCE in LDS (one batch of 32 global loads per thread) and all passes (maximum, the two
// radix selects, the Gumbel race: up to 11 sweeps) read it from there; the loop form re-reads the row from L2 in every sweep, one
// dependent 4-byte load at a time (the streamer path paid ~80 us per token for it).  Same values, same order-independent sums.
// (Keeping the 32 values in registers spills at 1024 threads per workgroup: 128 VGPRs.)
#define VZ_FOR_LOGITS(BODY)                                                                                          \
    if (REG) {                                                                                                       \
        _Pragma("unroll 8") for (int k = tid; k < cols; k += 1024) { const float x = lx[k]; BODY }                    \
    } else {                                                                                                         \
        for (int k = tid; k < cols; k += 1024) { const float x = lr[k] / temperature; BODY }                         \
    }

template <bool REG>
__global__ __launch_bounds__(1024) void sample_kernel(const float* __restrict__ logits, int cols, float temperature, int top_k,
                                                      float top_p, const unsigned* __restrict__ seed, const int* __restrict__ ctr,
                                                      int ctr_add, int* __restrict__ ids, int* __restrict__ pos, int* __restrict__ slot,
                                                      int* __restrict__ len, int* __restrict__ out_ids, int out_stride,
                                                      const int* __restrict__ step, int max_ctx, int rope_max, int* __restrict__ ring,
                                                      int ring_n) {
    __shared__ SampleShared sh;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* lr = logits + (size_t)row * cols;

    extern __shared__ float lx[];          // REG: cols floats
    if (REG) {
        float t[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { const int k = tid + i * 1024; t[i] = k < cols ? lr[k] : 0.f; }
#pragma unroll
        for (int i = 0; i < 32; ++i) { const int k = tid + i * 1024; if (k < cols) lx[k] = t[i] / temperature; }
        __syncthreads();
    }

    // ---- 1. max of the tempered logits ----
    float m = -INFINITY;
    VZ_FOR_LOGITS({ m = fmaxf(m, x); })
    m = block_max(m, sh);

    // ---- 2. top-k threshold key (0 = keep everything) ----
    unsigned kth = 0;
    if (top_k > 0 && top_k < cols) {
        unsigned prefix = 0, pmask = 0;
        unsigned long long want = (unsigned long long)top_k;          // rank from the top still to be found
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) sh.hist[tid] = 0;
            __syncthreads();
            VZ_FOR_LOGITS({
                const unsigned key = fkey(x);
                if ((key & pmask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255], 1u);
            })
            __syncthreads();
            if (tid == 0) {
                unsigned long long cum = 0;
                int b = 255;
                for (; b > 0; --b) { if (cum + sh.hist[b] >= want) break; cum += sh.hist[b]; }
                sh.sel = (unsigned)b; sh.carry = want - cum;
            }
            __syncthreads();
            prefix |= sh.sel << shift; pmask |= 255u << shift; want = sh.carry;
            __syncthreads();
        }
        kth = prefix;
    }

    // ---- 3. top-p threshold key ----
    unsigned pth = 0;
    if (top_p < 1.0f) {
        unsigned long long z = 0;
        VZ_FOR_LOGITS({
            if (fkey(x) >= kth) z += (unsigned long long)(expf(x - m) * 1099511627776.0f);
        })
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) z += __shfl_xor(z, o, 64);
        if ((tid & 63) == 0) sh.redm[tid >> 6] = z;
        __syncthreads();
        z = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) z += sh.redm[w];
        __syncthreads();
        // tokens whose cumulative mass (ascending) is <= (1 - top_p) * Z go; the first one beyond that stays
        const unsigned long long drop = (unsigned long long)((1.0 - (double)top_p) * (double)z);
        unsigned prefix = 0, pmask = 0;
        unsigned long long below = 0;                                  // mass of keys strictly below the current prefix range
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) sh.mass[tid] = 0;
            __syncthreads();
            VZ_FOR_LOGITS({
                const unsigned key = fkey(x);
                if (key >= kth && (key & pmask) == prefix)
                    atomicAdd(&sh.mass[(key >> shift) & 255], (unsigned long long)(expf(x - m) * 1099511627776.0f));
            })
            __syncthreads();
            if (tid == 0) {
                unsigned long long cum = below;
                int b = 0;
                for (; b < 255; ++b) { if (cum + sh.mass[b] > drop) break; cum += sh.mass[b]; }
                sh.sel = (unsigned)b; sh.carry = cum;
            }
            __syncthreads();
            prefix |= sh.sel << shift; pmask |= 255u << shift; below = sh.carry;
            __syncthreads();
        }
        pth = prefix;
    }
    const unsigned keep = kth > pth ? kth : pth;

    // ---- 4. Gumbel race over the kept tokens ----
    const unsigned k0 = seed[0], k1 = seed[1];
    const unsigned c2 = (unsigned)(ctr[0] + ctr_add);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    VZ_FOR_LOGITS({
        if (fkey(x) >= keep) {
            const unsigned w = philox_word0((unsigned)k, (unsigned)row, c2, 0x565au, k0, k1);
            const float u = ((float)(w >> 9) + 0.5f) * 1.1920928955078125e-07f;       // 2^-23; exact in fp32, inside (0, 1)
            const float v = x - logf(-logf(u));
            if (v > bv || (v == bv && k < bi)) { bv = v; bi = k; }
        }
    })
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sh.redf[tid >> 6] = bv; sh.redi[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sh.redf[w] > bv || (sh.redf[w] == bv && sh.redi[w] < bi)) { bv = sh.redf[w]; bi = sh.redi[w]; }
        if (bi == 0x7fffffff) bi = 0;     // all-NaN row
        ids[row] = bi;
        if (out_ids) out_ids[(size_t)row * out_stride + (step ? *step : 0)] = bi;
        if (ring) ring[(size_t)row * ring_n + (c2 % (unsigned)ring_n)] = bi;          // host-visible ring, slot = draw counter
        // saturating advance (a parked row of a continuous batch steps for ever inside its own cache row)
        if (pos && pos[row] + 1 < rope_max) pos[row] += 1;
        if (len && len[row] < max_ctx) { len[row] += 1; if (slot) slot[row] += 1; }
    }
}

}  // namespace

static bool g_sample_lds_ok = false;
// dynamic-LDS limit of the staged kernel (128 KiB for a 32768-wide row), set once and never inside a stream capture
int vz_init_sampling_kernels() {
    static VzDeviceOnce once;
    if (!vz_device_first(once)) return VZ_OK;
    VZ_CHECK_HIP(hipFuncSetAttribute((const void*)sample_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * (int)sizeof(float)));
    g_sample_lds_ok = true;
    return VZ_OK;
}

int vz_launch_sample(const float* logits, int rows, int cols, float temperature, int top_k, float top_p, const unsigned* seed,
                     const int* ctr, int ctr_add, int* ids, int* pos, int* slot, int* len, int* out_ids, int out_stride,
                     const int* step, int max_ctx, int rope_max, int* ring, int ring_n, hipStream_t s) {
    VZ_CHECK_ARG(logits && ids && seed && ctr && rows > 0 && cols > 0 && (!ring || ring_n > 0), "sample: bad argument");
    VZ_CHECK_ARG(temperature > 0.f && top_p > 0.f, "sample: temperature %g and top_p %g must be positive (temperature 0 = greedy: use argmax)", (double)temperature, (double)top_p);
    { int r = vz_init_sampling_kernels(); if (r) return r; }
    if (cols <= 32768 && g_sample_lds_ok)
        hipLaunchKernelGGL(sample_kernel<true>, dim3(rows), dim3(1024), (size_t)cols * sizeof(float), s, logits, cols, temperature, top_k, top_p, seed, ctr, ctr_add, ids, pos,
                           slot, len, out_ids, out_stride, step, max_ctx, rope_max, ring, ring_n);
    else
        hipLaunchKernelGGL(sample_kernel<false>, dim3(rows), dim3(1024), 0, s, logits, cols, temperature, top_k, top_p, seed, ctr, ctr_add, ids, pos,
                           slot, len, out_ids, out_stride, step, max_ctx, rope_max, ring, ring_n);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
