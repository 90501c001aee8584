// Decode-step attention for Zephyr-7B (gfx950), one launch per layer:
//   RoPE on the new token's Q and K (rotate-half, hf:models/mistral/modeling_mistral.py:51-81)
//   + append of K/V to the cache + softmax(q K^T) V over the cache (hf:...:139-178, GQA, sliding window)
//   + the combine of the context splits, all in ONE kernel.
//
// HBM/latency-bound: per layer and token the kernel streams the KV cache once (2 * 8 * ctx * 256 B = 8.4 MB
// at ctx 2048) and sits on the critical path of every decoded token, so the design minimises the dependent
// chain rather than the byte count:
//   * grid (nsplit, Hkv, B): a workgroup owns one KV head and one slice of the context and serves the 4 query
//     heads of that KV head from a single pass over K and V (GQA without repeat_kv, K/V read once, not 4x);
//   * the K and V rows of the slice are requested FIRST (16-byte loads straight to VGPRs, 16 lanes per
//     256-byte row, up to 8+8 rows per lane in flight); the RoPE of the new token runs under that latency;
//   * the per-split partials go to HBM with write-through (sc1) stores, one relaxed agent-scope ticket per
//     workgroup, and the LAST workgroup to arrive for a (slot, kv head) merges them with sc1 loads - no
//     release/acquire cache maintenance on the critical path (CDNA4 guide section 6 G16, valid form "sc1 payload +
//     drained ticket, last arriver told by the value its add returned"); placement independent.
// Every per-step quantity (position, cache slot) is read from device memory, so one captured hipGraph
// replays for every token.
#include "attn_decode_body.h"

namespace {

using namespace attn_dec;

// the body (attn_decode_body.h) is shared with the persistent decode-token kernel (decode_persist.hip)
__global__ __launch_bounds__(256) void attn_decode_fused_kernel(FusedParams p) {
    __shared__ Shared sm;
    const bf16_t* row = p.qkv + (size_t)blockIdx.z * (p.Hq + 2 * p.Hkv) * D;
    (void)body<false>(p, row, blockIdx.x, blockIdx.y, blockIdx.z, threadIdx.x, true, sm);
}

}  // namespace

int vz_launch_attn_decode_fused(const AttnDecodeFusedArgs& a, hipStream_t s) {
    VZ_CHECK_ARG(a.D == D && a.Hq == a.Hkv * G, "attn_decode_fused: needs head_dim 128 and 4 query heads per KV head");
    VZ_CHECK_ARG(a.nsplit >= 1 && a.nsplit <= 64 && a.qkv && a.kc && a.vc && a.o && a.part && a.ticket, "attn_decode_fused: bad argument");
    FusedParams p;
    p.qkv = a.qkv; p.kc = a.kc; p.vc = a.vc; p.o = a.o; p.part = a.part; p.ticket = a.ticket;
    p.cosT = a.cosT; p.sinT = a.sinT; p.pos = a.pos; p.slot = a.slot;
    p.B = a.B; p.Hq = a.Hq; p.Hkv = a.Hkv; p.max_ctx = a.max_ctx; p.nsplit = a.nsplit; p.window = a.window; p.scale = a.scale;
    hipLaunchKernelGGL(attn_decode_fused_kernel, dim3(a.nsplit, a.Hkv, a.B), dim3(256), 0, s, p);
    VZ_LAUNCH_CHECK();
    return VZ_OK;
}
