// Micro-benchmark (round 2): what bounds a weight-streaming kernel's read rate on MI355X?  One 16-KiB unit = 16 wave-instructions of 1 KiB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/stream_bench tools/micro/stream_bench.hip ; ./tools/micro/stream_bench
// Variants: workgroups x waves, units in flight per wave (1 = 16 KiB, 2 = 32 KiB), non-temporal, a workgroup barrier per unit, dynamic LDS
// (forces one workgroup per CU), 16 MFMAs per unit on the loaded data, and the unit ORDER: 0 = unit u of the launch at u x 16 KiB taken by wave
// u mod n_waves (the GEMV's order), 1 = every wave walks its own contiguous region (the tiled weight stream's order).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
struct P { const char* buf; size_t units; int nt, barrier, mfma, order, depth; unsigned* out; const char* x; int xmode; };

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream_kernel(P p) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t gw = (size_t)blockIdx.x * nw + wave, tw = (size_t)gridDim.x * nw;
    const size_t per = (p.units + tw - 1) / tw;
    u32x4 q[DEPTH][16];
    unsigned acc = 0;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    auto unit_addr = [&](size_t i) -> const char* {       // i-th unit of this wave
        const size_t u = p.order ? gw * per + i : gw + i * tw;
        return p.buf + (u < p.units ? u : p.units - 1) * 16384 + lane * 16;
    };
    auto issue = [&](int d, size_t i) {
        const char* a = unit_addr(i);
#pragma unroll
        for (int k = 0; k < 16; ++k) q[d][k] = p.nt ? __builtin_nontemporal_load((const u32x4*)(a + k * 1024)) : *(const u32x4*)(a + k * 1024);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d, d);
    for (size_t i = 0; i < per; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            u32x4 xr[8];
            if (p.xmode) {      // gemm_wide's activation staging, loads FIRST (older than the refills below): 8 wave-instructions from an L2-resident buffer
                const size_t ko = ((i + d) & 7) * 1024;        // the chunk's k offset (bytes) inside an 8 KiB row
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = wave + 8 * k;                // fragment index 0..63
                    const char* src = (p.xmode & 3) == 1 ? p.x + (size_t)((q & 3) * 16 + (lane & 15)) * 8192 + ko + (q >> 3) * 128 + ((q >> 2) & 1) * 16 + (lane >> 4) * 32
                                                         : p.x + (size_t)q * 8192 + ko + lane * 16;
                    xr[k] = *(const u32x4*)src;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p.mfma) {
#pragma unroll
                for (int k = 0; k < 16; ++k) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q[d][k]), __builtin_bit_cast(bf16x8, q[d][k ^ 1]), c, 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc ^= q[d][k][0] ^ q[d][k][3];
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // refill unconditionally (past the end: the last unit again) so that hipcc counts its waits
                const size_t nx = i + d + DEPTH < per ? i + d + DEPTH : per - 1;
                issue(d, nx);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (p.xmode) {      // -> LDS -> barrier (the 16 refills stay in flight: vmcnt(16)); xmode & 4: no barrier / LDS read
#pragma unroll
                for (int k = 0; k < 8; ++k) *(u32x4*)(smem + ((size_t)(wave + 8 * k) * 64 + ((p.xmode & 3) == 1 ? lane : ((lane * 5) & 63))) * 16) = xr[k];
                if (!(p.xmode & 4)) {
                    __syncthreads();
                    acc ^= *(const unsigned*)(smem + lane * 4);
                }
            }
            if (p.barrier) __syncthreads();
        }
    }
    if (smem[0] == 77) acc ^= 1;
    p.out[gw] = acc ^ __float_as_uint(c[0]);
}

int main() {
    const size_t bytes = (size_t)235 << 20, units = bytes / 16384;
    const int NB = 4;
    char* buf[NB];
    for (int i = 0; i < NB; ++i) { hipMalloc((void**)&buf[i], bytes); hipMemset(buf[i], i + 1, bytes); }
    unsigned* out; hipMalloc((void**)&out, 1 << 22);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct V { int wgs, nw, depth, nt, barrier, lds, mfma, order, xmode; };
    char* xbuf; hipMalloc((void**)&xbuf, 64 * 8192); hipMemset(xbuf, 3, 64 * 8192);
    std::vector<V> vs;
    for (int order = 1; order < 2; ++order) {
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 0});
        vs.push_back({224, 8, 1, 0, 1, 131072, 1, order, 0});
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 1}); // + scattered activation staging, barrier
        vs.push_back({224, 8, 1, 0, 0, 131072, 1, order, 1});
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 2}); // + coalesced activation staging, barrier
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 5}); // scattered loads + LDS write, no barrier
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 6}); // coalesced loads + LDS write, no barrier
        vs.push_back({224, 8, 1, 0, 0, 131072, 0, order, 0});
    }
    hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%6s %3s %5s %2s %3s %6s %4s %5s %5s   %8s %8s\n", "wgs", "nw", "depth", "nt", "bar", "lds", "mfma", "order", "xmode", "us", "TB/s");
    for (const V& v : vs) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            for (int it = 0; it < 8; ++it) {
                P p{buf[it % NB], units, v.nt, v.barrier, v.mfma, v.order, v.depth, out, xbuf, v.xmode};
                if (v.depth == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(v.wgs), dim3(v.nw * 64), v.lds, 0, p);
                else hipLaunchKernelGGL(stream_kernel<2>, dim3(v.wgs), dim3(v.nw * 64), v.lds, 0, p);
            }
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms / 8 < best) best = ms / 8;
        }
        printf("%6d %3d %5d %2d %3d %6d %4d %5d %5d   %8.1f %8.2f\n", v.wgs, v.nw, v.depth, v.nt, v.barrier, v.lds, v.mfma, v.order, v.xmode, best * 1e3, bytes / (best * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
