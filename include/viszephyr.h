/*
 * viszephyr.h - C ABI of libviszephyr_hip.so: the MI355X (gfx950) implementation of the
 * Vision-Zephyr forward/generate hot path.
 *
 * The reference (baohuyvanba/Vision-Zephyr) is 100 % Python and has no FFI layer; its boundary
 * for this path is the Python class API of
 *     vis_zephyr/model/language_model/vis_zephyr.py:28-170   (VisZephyrForCausalLM.forward/generate)
 *     vis_zephyr/model/vis_zephyr_arch.py:120-333            (encode_images, prepare_inputs_labels_for_multimodal)
 * whose arithmetic it delegates to HF transformers / torch.nn.  This header is what a maintainer
 * binds (ctypes stub in INTEGRATION.md) to replace that arithmetic: plain pointers and sizes, no
 * torch types, int status codes, no exceptions across the boundary.  Every pointer named `d_*`
 * is a DEVICE pointer borrowed for the duration of the call (weights: for the engine's lifetime);
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  bf16 tensors are uint16_t.
 *
 * Two layers:
 *   1. operator entry points  (vz_op_*): one hand-written kernel each; used by the parity tests
 *      and by the engine itself.
 *   2. engine entry points    (vz_engine_* / vz_clip_* / vz_qformer / vz_llm_*): own workspace and
 *      the KV cache, sequence the kernels of one stage on the caller's stream.
 */
#ifndef VISZEPHYR_H
#define VISZEPHYR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VZ_OK 0
#define VZ_ERR_ARG 1          /* bad argument / unsupported shape  -> Python ValueError      */
#define VZ_ERR_HIP 2          /* HIP runtime error                 -> Python RuntimeError    */
#define VZ_ERR_STATE 3        /* call order / missing weights      -> Python RuntimeError    */
#define VZ_ERR_UNSUPPORTED 4  /* reference-unreachable feature     -> NotImplementedError    */

/* values of the async error word (vz_engine_async_error / vz_op_async_error): which bounded device-side wait expired */
/* (1 was the hand-off of the one-launch batch-1 attention half, removed in round 2: it never beat the three kernels) */
#define VZ_ASYNC_STREAMK 2    /* stream-K fix-up of the 256^2 GEMM: the tile was written as NaN, never as a sum of stale slots */
#define VZ_ASYNC_ONESHOT 4    /* a peer's vector never arrived in the one-shot all-reduce (comm_oneshot.hip): the output was poisoned with NaN */
#define VZ_ASYNC_ATTN_O 5     /* the O-projection workgroups of the fused attention + O launch (attn_o_fused.hip) never saw the mergers arrive: x poisoned with NaN */
#define VZ_ASYNC_PERSIST 3    /* a phase hand-off of the persistent decode-token kernel (decode_persist.hip) expired: that token's logits are garbage */

typedef void* vz_stream;
typedef struct vz_engine vz_engine;

/* last error message of the calling thread ("" if none) */
const char* vz_last_error(void);
/* ABI version; bumped on any signature change */
int vz_abi_version(void);
/* name of the code object's target ("gfx950") */
const char* vz_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * 1. Operator entry points
 * ------------------------------------------------------------------------------------------ */

/* epilogue activation ids */
#define VZ_ACT_NONE 0
#define VZ_ACT_QUICK_GELU 1   /* x*sigmoid(1.702x)  hf:activations.py:117-123 (CLIPMLP)            */
#define VZ_ACT_GELU_ERF 2     /* exact GELU         torch nn.GELU (Q-Former FFN, ref builder.py:29) */
#define VZ_ACT_SWIGLU 3       /* silu(gate)*up, W rows interleaved [16 gate | 16 up]; out width N/2
                                 hf:models/mistral/modeling_mistral.py:35-48 (MistralMLP)          */

/* C[M,N'] = epi(A[M,K] . W[N,K]^T): every nn.Linear on the path
 * (hf:models/clip/modeling_clip.py:280-350, hf:models/mistral/modeling_mistral.py:35-48,122-178,
 *  torch MultiheadAttention in/out projections ref:vis_zephyr/model/multimodal_projector/builder.py:16-32).
 * bf16 A/W, fp32 accumulate (MFMA), bias fp32 [N] or NULL, residual bf16 [M,ldr] or NULL (may alias C),
 * out bf16 (out_fp32=0) or fp32.  K % 64 == 0.  M <= 8 routes to the weight-streaming GEMV. */
int vz_op_linear(const void* d_A, int lda, const void* d_W, int ldw, void* d_C, int ldc,
                 int M, int N, int K, const float* d_bias, const void* d_residual, int ldr,
                 int act, int out_fp32, vz_stream stream);
/* W8A16 weight stream for M <= 64 (decode batches; fused RMSNorm: M <= 16): W as OCP e4m3 rows [N, ldw bytes] + one fp32 power-of-two scale per row
 * (vz_hip/quant.py: 2^e * fp8 is exactly a bf16 number, so the bf16 GEMMs of the prefill run on the same weights).
 * Same epilogues as vz_op_linear; K % 1024 == 0; d_norm_w != NULL fuses the RMSNorm of x into the staging. */
int vz_op_linear_fp8(const void* d_A, int lda, const void* d_W8, int ldw, const float* d_wscale, void* d_C, int ldc,
                     int M, int N, int K, const float* d_bias, const void* d_residual, int ldr, int act, int out_fp32,
                     const float* d_norm_w, float norm_eps, vz_stream stream);
/* FP8 MFMA linear (config 5's prefill on a weight_fp8 engine): e4m3 activations x e4m3 weights on v_mfma_scale_f32_16x16x128_f8f6f4,
 * one power-of-two scale per activation row and per weight row applied to the fp32 sum, then vz_op_linear's epilogues.
 * vz_op_quant_rows_fp8 is the activation quantiser: scale[r] = 2^e with e the smallest integer such that max|x_row| <= 448 * 2^e,
 * q = e4m3(x * 2^-e) round-to-nearest-even - the weights' quantiser (vz_hip/quant.py::quantize_rows), byte for byte.  K % 128 == 0. */
/* RMSNorm + that quantiser in one launch (the bytes and scales of vz_op_rmsnorm followed by vz_op_quant_rows_fp8); cols <= 5120 */
int vz_op_rmsnorm_quant_fp8(const void* d_x, int ldx, const float* d_w, float eps, void* d_q, int ldq, float* d_scale, int rows, int cols, vz_stream stream);
int vz_op_quant_rows_fp8(const void* d_x, int ldx, void* d_q, int ldq, float* d_scale, int rows, int K, vz_stream stream);
int vz_op_linear_fp8_mfma(const void* d_A8, int lda, const float* d_ascale, const void* d_W8, int ldw, const float* d_wscale, void* d_C, int ldc,
                          int M, int N, int K, const float* d_bias, const void* d_residual, int ldr, int act, int out_fp32, vz_stream stream);
/* x -> bf16(norm_w * x * rsqrt(mean(x^2) + eps)) fused into the staging of the weight-stream kernels (decode: QKV, gate-up,
 * lm_head; hf:models/mistral/modeling_mistral.py:182-199), then vz_op_linear's contract without bias.  1 <= M <= 16. */
int vz_op_linear_rmsnorm(const void* d_A, int lda, const float* d_norm_w, float norm_eps, const void* d_W, int ldw, void* d_C, int ldc,
                         int M, int N, int K, const void* d_residual, int ldr, int act, int out_fp32, vz_stream stream);
/* Fragment-tiled copy of a dense bf16 weight [N, K] (N % 16 == 0, K % 64 == 0) for the 2..64-row MFMA weight stream: the 16 bytes
 * W[16 G + r][64 s + 16 g + 8 j .. + 7] move to chunk ((G * K/64 + s) * 2 + j) * 64 + (16 g + r), so that every wave-instruction of the
 * stream reads 1 KiB contiguous.  An engine uses the copy registered as "<weight name>t" (same element count) for its decode
 * linears; vz_op_linear_tiled is the same launch at op level (results bit-identical to the row-major stream, impl 3). */
int vz_op_tile_weights(const void* d_W, int N, int K, int ldw, void* d_Wt, vz_stream stream);
int vz_op_linear_tiled(const void* d_A, int lda, const void* d_W, const void* d_Wt, int ldw, void* d_C, int ldc, int M, int N, int K,
                       const float* d_bias, const void* d_residual, int ldr, int act, int out_fp32, const float* d_norm_w,
                       float norm_eps, vz_stream stream);
/* W8A16 form for 17..64 rows (ABI 10; SURVEY config 5: batched decode of the e4m3-weight engine - the reference's analogue is
 * load_8bit / load_4bit, ref:vis_zephyr/model/builder.py:33-43, applied to every generate call of the eval loop,
 * ref:vis_zephyr/eval/eval_vqa.py:176-199).  vz_op_tile_weights_fp8: e4m3 rows [N][ldw bytes] -> fragment order
 * W8t[((G * K/64 + s) * 64 + 16 g + r) * 16 + i] = W8[16 G + r][64 s + 16 g + i] (one 1-KiB wave-instruction per 64-k step and 16-row
 * group).  vz_op_linear_tiled_fp8: C = epi(A . (2^e_row * W8)^T) on that copy, bf16 activations, 17 <= M <= 64, N % 128 == 0,
 * K % 1024 == 0; an engine registers the copy as "<e4m3 name>t" (dtype 2) and its 17..64-row decode steps stream it. */
int vz_op_tile_weights_fp8(const void* d_W8, int N, int K, int ldw, void* d_W8t, vz_stream stream);
int vz_op_linear_tiled_fp8(const void* d_A, int lda, const void* d_W8t, const float* d_wscale, void* d_C, int ldc, int M, int N, int K,
                           const float* d_bias, const void* d_residual, int ldr, int act, int out_fp32, vz_stream stream);
/* same contract, forcing one implementation (tests): impl 0 = 128^2 MFMA tile GEMM, 1 = GEMV (M <= 8), 2 = 256^2 tile GEMM,
 * 3 = MFMA weight stream for 2 <= M <= 64 (batched decode), 4 = the 128^2 tile GEMM with the finer split-K of the 17..64-row decode route */
int vz_op_linear_impl(int impl, const void* d_A, int lda, const void* d_W, int ldw, void* d_C, int ldc,
                      int M, int N, int K, const float* d_bias, const void* d_residual, int ldr,
                      int act, int out_fp32, vz_stream stream);

/* Causal-LM loss of forward(labels=...) (ref:vis_zephyr/model/language_model/vis_zephyr.py:51-98 hands labels to MistralForCausalLM.forward; hf:loss/loss_utils.py ForCausalLMLoss):
 * fp32 logits [B, S, V], int32 labels [B, S] (-100 = ignored), row (b, s) scored against labels[b][s + 1], mean over the valid targets.
 * d_loss_rows: B * S floats of scratch; d_out[0] = loss (NaN when no target is valid, as torch's mean), d_out[1] = number of valid targets. */
int vz_op_causal_lm_loss(const float* d_logits, int B, int S, int V, const int* d_labels, float* d_loss_rows, float* d_out, vz_stream stream);

/* y = LayerNorm(x) (torch.nn.LayerNorm; CLIP hf:...modeling_clip.py:353-384, Q-Former builder.py:14-27,68-70) */
int vz_op_layernorm(const void* d_x, int ldx, void* d_y, int ldy, const float* d_w, const float* d_b,
                    int rows, int cols, float eps, vz_stream stream);
/* y = w * x * rsqrt(mean(x^2)+eps)  (MistralRMSNorm hf:models/mistral/modeling_mistral.py:182-199) */
int vz_op_rmsnorm(const void* d_x, int ldx, void* d_y, int ldy, const float* d_w,
                  int rows, int cols, float eps, vz_stream stream);

/* softmax(scale * Q K^T [+causal/window mask]) V, fp32 softmax, GQA by head grouping
 * (hf:models/clip/modeling_clip.py:259-277; hf:models/mistral/modeling_mistral.py:84-119;
 *  torch MultiheadAttention).  Element strides (batch, seq, head) per tensor; head_dim in {64,128,512}.
 * causal: query i of batch b sits at absolute position q_pos0 + i and sees keys j <= position,
 * j > position - window (window <= 0: unlimited).  d_kv_len (int32 [B]) or NULL (= Sk) bounds the
 * valid keys of each batch row. */
int vz_op_attention(const void* d_q, const void* d_k, const void* d_v, void* d_o,
                    int B, int Sq, int Sk, int Hq, int Hkv, int head_dim,
                    long q_bs, long q_ss, long q_hs, long k_bs, long k_ss, long k_hs,
                    long v_bs, long v_ss, long v_hs, long o_bs, long o_ss, long o_hs,
                    float scale, int causal, int q_pos0, int window, const int* d_kv_len,
                    vz_stream stream);

/* The same operator given an fp32 workspace of `workspace_floats` elements: a launch with Sq <= 64 un-masked query rows,
 * head_dim 512 and many keys (the Q-Former cross-attention of ref:vis_zephyr/model/multimodal_projector/builder.py:34-39,
 * 32 queries x 576 visual tokens per tile) gives every 96 keys their own workgroup and merges the partial softmaxes in a second
 * kernel (the split depends on Sk only: a row's result is independent of B).  Needs B*Hq*ceil(Sk/96)*Sq*516 floats; with less
 * (or any other shape) it runs as vz_op_attention. */
int vz_op_attention_split(const void* d_q, const void* d_k, const void* d_v, void* d_o,
                          int B, int Sq, int Sk, int Hq, int Hkv, int head_dim,
                          long q_bs, long q_ss, long q_hs, long k_bs, long k_ss, long k_hs,
                          long v_bs, long v_ss, long v_hs, long o_bs, long o_ss, long o_hs,
                          float scale, int causal, int q_pos0, int window, const int* d_kv_len,
                          float* d_workspace, long workspace_floats, vz_stream stream);

/* Attention BACKWARD, tile-resident (attn_bwd_flash.hip): dQ, dK, dV of softmax(scale Q K^T + mask) V without an Sq x Sk tensor in memory - the
 * form the reference trains through (ref:vis_zephyr/train/zephyr_flash_attn_monkey_patch.py:100-124; masks as vz_op_attention: causal,
 * sliding window, per-sample key count).  head_dim 128.  q, dO, dq [B,Sq,Hq,D] bf16; k, v [B,Hkv,Sk,D] bf16; dk, dv [B,Hkv,Sk,D] fp32
 * (dkv_fp32 = 1) or bf16, already summed over the query heads of a KV head.  d_ws: >= 2 * B * Hq * Sq + 64 floats. */
int vz_op_attention_bwd(const void* d_q, const void* d_k, const void* d_v, const void* d_dO, void* d_dq, void* d_dk, void* d_dv, int dkv_fp32,
                        int B, int Sq, int Sk, int Hq, int Hkv, int head_dim, float scale, int causal, int window, const int* d_kv_len,
                        float* d_ws, long ws_floats, vz_stream stream);
/* Two helpers of the training step (train.hip) at op level: vz_op_transpose dst[c][r] = src[r][c] (bf16 [R,C] -> [C,R]; 64 x 64 tiles through the
 * hardware transposing LDS read when R, C and the leading dimensions are multiples of 8 and the bases 16-byte aligned), vz_op_colsum
 * d_out[c] += sum_r y[r][c] (fp32, the bias gradients; fixed summation order) with d_part >= vz_op_colsum_groups(rows) * cols floats. */
int vz_op_transpose(const void* d_src, long src_ld, void* d_dst, long dst_ld, int R, int C, vz_stream stream);
int vz_op_colsum(const void* d_y, int ld, long rows, int cols, float* d_part, long part_floats, float* d_out, vz_stream stream);
int vz_op_colsum_groups(long rows);
/* RoPE (rotate-half, hf:models/mistral/modeling_mistral.py:51-81) on the Q and K heads of a fused QKV row
 * [B*S, (Hq+2Hkv)*D] + append of K/V to the cache [B][Hkv][max_ctx][D].  d_pos / d_slot: int32 [B*S] position
 * id and cache slot of every token (slot < 0: token not cached).  d_q_out bf16 [B*S,Hq,D].  D = 128. */
int vz_op_rope_kv(const void* d_qkv, int ld, void* d_q_out, void* d_kcache, void* d_vcache, const float* d_cos,
                  const float* d_sin, const int* d_pos, const int* d_slot, int B, int S, int Hq, int Hkv, int head_dim,
                  int max_ctx, vz_stream stream);

/* one query token per slot against the KV cache (decode step of hf:...modeling_mistral.py:139-178).
 * d_q / d_o bf16 [B,Hq,D]; caches [B][Hkv][max_ctx][D]; d_ctx_len int32 [B] = keys visible (device memory, so a
 * captured graph replays for every step); d_workspace fp32 [B*Hq*nsplit*(D+2)]. */
int vz_op_attention_decode(const void* d_q, const void* d_kcache, const void* d_vcache, void* d_o, float* d_workspace,
                           int B, int Hq, int Hkv, int head_dim, int max_ctx, int nsplit, int window, float scale,
                           const int* d_ctx_len, vz_stream stream);

/* The decode step's attention as the engine runs it: RoPE on the new token's Q/K, append of K/V to the cache,
 * attention over the cache (4 query heads per KV head served from one pass over K/V) and the merge of the
 * nsplit context slices, in ONE launch.  d_qkv bf16 [B,(Hq+2Hkv)*D] (un-rotated projection of the new token);
 * d_pos / d_slot int32 [B] (position id, cache slot = tokens already cached); d_ticket uint32 [B*Hkv], zeroed
 * once by the caller; d_workspace fp32 [B*Hkv*nsplit*(4*D+32)].  Same result as vz_op_rope_kv followed by
 * vz_op_attention_decode. */
int vz_op_attention_decode_fused(const void* d_qkv, void* d_kcache, void* d_vcache, void* d_o, float* d_workspace,
                                 unsigned* d_ticket, const float* d_cos, const float* d_sin, const int* d_pos,
                                 const int* d_slot, int B, int Hq, int Hkv, int head_dim, int max_ctx, int nsplit,
                                 int window, float scale, vz_stream stream);

/* ------------------------------------------------------------------------------------------
 * 2. Engine
 * ------------------------------------------------------------------------------------------ */
typedef struct vz_config {
    /* Zephyr-7B-beta (ref:checkpoints/vis-zephyr-7b-v1-pretrain/config.json:10-38) */
    int hidden, inter, n_layers, n_heads, n_kv_heads, head_dim, vocab;
    float rms_eps, rope_theta;
    int sliding_window;
    /* CLIP ViT-L/14-336 (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:21) */
    int clip_hidden, clip_inter, clip_layers, clip_heads, clip_image, clip_patch;
    float clip_eps;
    /* Q-Former (ref:vis_zephyr/model/multimodal_projector/builder.py:49-70) */
    int qf_queries, qf_blocks, qf_heads, qf_kv_dim;
    float qf_eps;
    /* fusion (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:63-64) */
    int fusion_groups, fusion_layers_per_group;
    /* capacity */
    int max_batch;        /* KV-cache slots                                  */
    int max_ctx;          /* tokens per slot                                 */
    int max_tiles;        /* tiles per vz_clip_fused_features / vz_qformer   */
    int max_text;         /* Lmax of the Q-Former text conditioning          */
    /* tensor parallelism over RCCL (1 = single GPU) */
    int tp_size, tp_rank;
    /* mm_vision_select_feature (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:66-73): 0 = 'patch' (drop CLS, 576 tokens per
     * tile, the shipped config), 1 = 'cls_patch' (keep it: 577 tokens per tile into the fusion and the Q-Former) */
    int clip_keep_cls;
    /* 1 = the decode-side linears of Zephyr (q|k|v, o, gate|up, down, lm_head at M <= 8) stream e4m3 copies of the weights
     * ("<name>8" u8 + "<name>s" fp32 row scales, registered beside the bf16 tensors); the bf16 tensors must then hold the
     * dequantised values so that prefill and decode run the same model (vz_hip/quant.py, SURVEY config 5) */
    int weight_fp8;
} vz_config;

int vz_engine_create(const vz_config* cfg, vz_engine** out);
int vz_engine_destroy(vz_engine* e);

/* Register one weight under its engine name (see vz_hip/weights.py for the packing from the
 * reference's state-dict keys).  dtype: 0 = bf16, 1 = fp32, 2 = u8 (e4m3 bytes).  The pointer is borrowed until destroy. */
int vz_engine_set_weight(vz_engine* e, const char* name, const void* d_ptr, int dtype, long n_elems);
/* check that every weight the configuration needs has been registered */
int vz_engine_finalize(vz_engine* e);
/* change the vocabulary of a live engine (tokens added to the tokenizer: ref:vis_zephyr/model/builder.py:141-153 grows
 * embed_tokens / lm_head by <im_patch>).  The caller then registers the new "llm.embed" / "llm.lm_head" tables and
 * finalizes again.  tp_size == 1 only. */
int vz_engine_resize_vocab(vz_engine* e, int new_vocab);
/* Tensor parallelism (vz_config.tp_size > 1): one process per GPU; q/k/v/gate/up column-parallel,
 * o/down row-parallel with an RCCL all-reduce of [B,S,hidden] bf16 after each (2 per layer), lm_head vocab-parallel with
 * an all-gather of the fp32 logits; the KV cache is sharded by KV head.  CLIP / Q-Former weights are replicated.
 * vz_comm_unique_id fills a 128-byte ncclUniqueId on one rank; the caller ships it to the other ranks (any transport) and
 * every rank calls vz_comm_init with it before its first prefill. */
int vz_comm_unique_id(char* out128);
int vz_comm_init(vz_engine* e, const char* id128);
/* Tile data parallelism (first row): every anyres tile is an independent unit through CLIP, fusion and
 * the Q-Former, so the host deals tile t to rank t mod tp_size, each rank encodes its tiles, and ONE all-gather of
 * bytes_per_rank bytes per rank (ceil(T / tp_size) x 32 x hidden bf16, zero-padded) hands every rank all visual tokens.
 * d_recv holds tp_size consecutive chunks in rank order.  At tp_size == 1 this is a device copy. */
int vz_tp_all_gather(vz_engine* e, const void* d_send, void* d_recv, size_t bytes_per_rank, vz_stream stream);

/* fp32 rotary tables [max_pos, head_dim/2] (rotate-half convention, hf:...modeling_mistral.py:51-81,262-317) */
int vz_engine_set_rope(vz_engine* e, const float* d_cos, const float* d_sin, int max_pos);

/* a8-a10: CLIPVisionTower.forward + feature_select + fusion
 * (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:58-117, gating_fusion.py:22-50).
 * d_images bf16 [T,3,336,336] -> d_out bf16 [T,576,5*clip_hidden].
 * d_hidden_dbg: NULL, or bf16 [(clip_layers+1),T,577,clip_hidden] receiving every hidden state (tests). */
int vz_clip_fused_features(vz_engine* e, const void* d_images, int T, void* d_out, void* d_hidden_dbg,
                           vz_stream stream);

/* a11: QFormer.forward (ref:vis_zephyr/model/multimodal_projector/builder.py:72-92).
 * d_feats bf16 [T,576,qf_kv_dim]; text conditioning per SAMPLE: d_text bf16 [n_samples,Lmax,hidden]
 * (rows past a sample's own length are the zero padding of ref:vis_zephyr/model/vis_zephyr_arch.py:178-189),
 * h_tile_sample[T] maps each tile to its sample (the reference expands the same text to every tile of a
 * sample, vis_zephyr_arch.py:174).  Lmax = 0: no text.  d_out bf16 [T,32,hidden].
 * Layout hint: when the blocks' "qf.<i>.ca_kv.w" (and ".b") tensors lie back to back in device memory, block-major - one
 * [qf_blocks * 2 * hidden, qf_kv_dim] allocation registered slice by slice - the cross-attention K|V projections of all blocks run
 * as ONE product (they read the same pre-normed features); any other placement computes them block by block, same results up to
 * the stream-K tail's fp32 re-association. */
int vz_qformer(vz_engine* e, const void* d_feats, int T, const void* d_text, int n_samples, int Lmax,
               const int* h_tile_sample, void* d_out, vz_stream stream);

/* a6/a7 data movement: out[r,:] = kind[r]==0 ? embed_tokens[idx[r]] : kind[r]==1 ? d_visual[idx[r]] : 0
 * (the row map itself is host index logic, ref:vis_zephyr/model/vis_zephyr_arch.py:214-333,476-530). */
int vz_embed_splice(vz_engine* e, const int* d_kind, const int* d_idx, int rows, const void* d_visual,
                    void* d_out, vz_stream stream);

/* a12 prefill: MistralForCausalLM.forward on inputs_embeds (hf:models/mistral/modeling_mistral.py:340-466).
 * d_embeds bf16 [B,S,hidden] right-padded, h_seqlens[B] valid lengths, d_pos int32 [B,S] position ids.
 * Fills KV slots 0..B-1 from position 0.  d_logits_all: NULL or fp32 [B,S,vocab] (forward() computes all
 * positions); d_logits_last: NULL or fp32 [B,vocab] (row seqlen-1 of each sample; what generate needs). */
int vz_llm_prefill(vz_engine* e, const void* d_embeds, int B, int S, const int* h_seqlens, const int* d_pos,
                   float* d_logits_all, float* d_logits_last, vz_stream stream);

/* a13 greedy decode.  vz_llm_decode_begin arms the device-side state after a prefill: the first input
 * token of every slot (d_first_ids int32 [B]), next position ids and context lengths (host int [B]).
 * vz_llm_decode_steps enqueues n steps with no host synchronisation: every step embeds the current token,
 * runs the 32 layers against the cache, takes argmax of the fp32 logits (first maximal index, as
 * torch.argmax) and feeds it to the next step.  d_out_ids int32 [B, n] receives the n new tokens;
 * d_logits_dbg NULL or fp32 [n,B,vocab]. */
int vz_llm_decode_begin(vz_engine* e, int B, const int* d_first_ids, const int* h_next_pos, const int* h_ctx_len,
                        vz_stream stream);
int vz_llm_decode_steps(vz_engine* e, int n, int* d_out_ids, float* d_logits_dbg, vz_stream stream);
/* vz_llm_decode_steps rejects (VZ_ERR_ARG) a call whose n steps would take a live row past max_ctx keys or past the rotary
 * tables; rows parked by vz_llm_decode_set_row(ctx_len 0) saturate on the device instead.
 *
 * Sampling tail (hf:generation/utils.py `_sample`, do_sample=True; ref:vis_zephyr/serve/cli.py:171-182 = temperature 0.2 +
 * HF's default top_k 50): after vz_llm_decode_sampling(enable=1) every decode step draws its token on the device -
 * logits / temperature, top-k, top-p, then a Gumbel race keyed by Philox4x32-10(seed; vocab index, row, draw counter) - inside
 * the same per-token hipGraph, so a streamer / stopping-criteria loop reads back 4 bytes per token.  `first_counter` is the
 * draw counter the next vz_llm_decode_begin starts from.  vz_op_sample: the same kernel on arbitrary fp32 logits rows
 * (the first token, drawn from the prefill logits with counter 0; tests).  Restated in oracle/sampling_oracle.py. */
int vz_llm_decode_sampling(vz_engine* e, int enable, float temperature, int top_k, float top_p, unsigned long long seed,
                           int first_counter);
/* Streamer / stopping-criteria loop (ref:vis_zephyr/serve/cli.py:155-182 passes a TextStreamer and a KeywordsStoppingCriteria:
 * one host callback per token): every step's tail also writes its token to ring[row * ring_n + (draw counter mod ring_n)], a
 * device-visible HOST buffer of ring_rows x ring_n ints, so the host keeps a step in flight and reads token t when the event behind
 * step t fires.  vz_llm_decode_steps refuses a decode batch with more rows than the ring holds (ABI 10: ring_rows). */
int vz_llm_decode_ring(vz_engine* e, int* host_visible_ring, int ring_n, int ring_rows);
int vz_op_sample(const float* d_logits, int rows, int cols, float temperature, int top_k, float top_p, unsigned long long seed,
                 int counter, int* d_ids, vz_stream stream);
/* how the last vz_llm_decode_steps ran: *graph = 1 when a captured hipGraph was replayed; *comm_in_graph = 1 when the RCCL
 * collectives of a tensor-parallel engine are part of that graph (0 = eager steps, e.g. after RCCL refused the capture) */
int vz_llm_decode_mode(vz_engine* e, int* graph, int* comm_in_graph);
/* Batch-1 decode on an MI355X (ABI 10): after vz_tune_set(28, 1), with one row, one GPU, bf16 weights and the Zephyr-7B geometry,
 * vz_llm_decode_steps runs every token as ONE resident grid (decode_persist.hip: the 161 launches of a step become phases with
 * in-launch hand-offs; bit-identical logits).  Opt-in: on MI355X it measured slower than the launch chain.  A phase hand-off that expires raises VZ_ASYNC_PERSIST (vz_engine_async_error).
 * vz_test_persist_poke is a test hook: *mode = 1 if the last steps ran that way; word >= 0 presets an arrival counter. */
int vz_test_persist_poke(vz_engine* e, int word, unsigned value, int* mode, vz_stream stream);
/* One-shot all-reduce of the tensor-parallel decode step (ABI 10; comm_oneshot.hip): every rank stores its bf16 vector into every
 * peer's receive area as 8-byte {two bf16, sequence tag} granules (system scope: over xGMI when the areas are peer-mapped) and sums the
 * N vectors it finds in its own area in rank order - one store-and-poll round instead of a ring.  `d_areas[q]` = rank q's receive area
 * (vz_op_oneshot_area_bytes(n_ranks, max_elems) bytes, zero-filled once; peer areas opened through hipIpc by the caller), `d_seq` = two
 * device words {sequence number = 1, ticket = 0} (a zero-filled area carries tag 0 = "nothing yet", so the numbering starts at 1); the launch advances the sequence number itself, so it replays
 * from a hipGraph.  An engine with tp_size > 1 uses it for its decode-step all-reduces after vz_comm_oneshot_attach (RCCL otherwise).
 * Tested in one process (N areas on one GPU, N concurrent launches); unmeasured over xGMI - no multi-GPU box was available. */
size_t vz_op_oneshot_area_bytes(int n_ranks, int max_elems);
int vz_op_allreduce_oneshot(void* const* d_areas, int rank, int n_ranks, int max_elems, const void* d_in, void* d_out, int n, unsigned* d_seq,
                            int* d_err, vz_stream stream);
/* test form: all n_ranks ranks as slices of one grid (one process, one GPU); d_in / d_out / d_seq are arrays of n_ranks pointers */
int vz_test_allreduce_oneshot_all(void* const* d_areas, int n_ranks, int max_elems, const void* const* d_in, void* const* d_out, int n,
                                  unsigned* const* d_seq, int* d_err, vz_stream stream);
/* engine side: this rank's receive area + sequence words (allocated on first call; out_area may be exported with hipIpcGetMemHandle),
 * then the N areas in rank order (own included) - from then on the decode step's [B, hidden] all-reduces take the one-shot kernel */
int vz_comm_oneshot_local(vz_engine* e, void** out_area, size_t* out_bytes);
int vz_comm_oneshot_attach(vz_engine* e, void* const* d_areas, int n_ranks);
/* profiling: s_memrealtime stamps (100 MHz) of workgroup 0's sync wave at the 12 phase edges of every layer of the last token */
int vz_prof_persist_stamps(vz_engine* e, unsigned long long* host_out, int n_layers);
/* batch-1 decode runs QKV GEMV + attention + O GEMV of a layer as ONE launch whose roles hand over through device-side counters;
 * every device-side wait is bounded and raises a word when it expires.  Reads and clears that word (blocking): *err != 0 = the
 * outputs since the previous call are invalid. */
int vz_engine_async_error(vz_engine* e, int* err);
/* The same word for op-level launches (vz_op_linear* whose 256^2 GEMM took the stream-K tail) on `stream`.  Stream-K state
 * (fp32 slots, arrival tickets, error word) exists once per (device, stream): launches that share it are stream-ordered.  A wait
 * that expires raises VZ_ASYNC_STREAMK, leaves the tickets untouched and poisons its tile with NaN; reading the error resets the
 * tickets of that stream. */
int vz_op_async_error(vz_stream stream, int* err);
/* TEST HOOK (tests/test_ops_gpu.py): overwrite the {arrive, ready} ticket pair of stream-K remainder tile `tile` on `stream`. */
int vz_test_corrupt_streamk(vz_stream stream, int tile, int arrive, int ready);
/* weight_fp8 engines: enable = 1 runs the Zephyr prefill linears as e4m3 x e4m3 on the scaled MFMA (inputs quantised per row by
 * vz_op_quant_rows_fp8's kernel, weights = the registered e4m3 copies); 0 (default) = bf16 MFMA on the dequantised bf16 tensors */
int vz_engine_prefill_fp8(vz_engine* e, int enable);
/* forget a registered weight (e.g. the e4m3 copy of a bf16 tensor that has been rewritten) */
int vz_engine_unset_weight(vz_engine* e, const char* name);
/* Continuous batching.  vz_llm_prefill_rows: vz_llm_prefill into KV-cache rows row0 .. row0+B-1;
 * vz_llm_decode_set_row: (re)arm one row of the running decode batch - next input token, rotary position, context length -
 * without touching the others (a finished row is parked with ctx_len 0 until a new request is prefilled into it). */
int vz_llm_prefill_rows(vz_engine* e, int row0, const void* d_embeds, int B, int S, const int* h_seqlens, const int* d_pos,
                        float* d_logits_all, float* d_logits_last, vz_stream stream);
int vz_llm_decode_set_row(vz_engine* e, int row, int token, int next_pos, int ctx_len, vz_stream stream);
/* Batched admissions: prefill several requests together into spare cache rows (row0 >= the running batch), then move each one's
 * first h_len[i] cache positions from row h_src[i] to the freed row h_dst[i] (every layer, K and V; stream-ordered).  Rows of
 * one call must not overlap (no destination equal to another move's source or destination). */
int vz_llm_kv_move_rows(vz_engine* e, int n, const int* h_src, const int* h_dst, const int* h_len, vz_stream stream);

/* ---- anyres preprocessing on the device (ref:vis_zephyr/model/multi_scale_process.py:70-171) ----
 * vz_op_resample_u8: Pillow's 8-bit LANCZOS `Image.resize` (horizontal pass, 8-bit intermediate, vertical pass) of an
 * [h, w, 3] u8 image to [h2, w2, 3], bit-exact.  bounds int32 [out, 2] = (first tap, tap count), coefs int32 [out, k] at 22
 * fractional bits, computed by the host as Pillow's precompute_coeffs / normalize_coeffs_8bpc do (vz_hip/preprocess.py).
 * d_tmp [h, w2, 3] is needed when both sizes change; equal sizes copy.
 * vz_op_anyres_tiles: the letterbox paste on black + row-major side x side crops + the global view in front + CLIP rescale /
 * normalise through a bf16 LUT [3, 256] -> bf16 [1 + grid_w * grid_h, 3, side, side], what the vision tower takes. */
int vz_op_resample_u8(const void* d_src, int h, int w, void* d_tmp, void* d_dst, int h2, int w2, const int* d_xbounds,
                      const int* d_xcoefs, int kx, const int* d_ybounds, const int* d_ycoefs, int ky, vz_stream stream);
int vz_op_anyres_tiles(const void* d_global, const void* d_resized, int nh, int nw, int paste_x, int paste_y, int grid_w,
                       int grid_h, int side, const void* d_lut, void* d_out, vz_stream stream);

/* ------------------------------------------------------------------------------------------
 * 3. Stage-1 pretrain step
 *
 * ref:vis_zephyr/train/train.py:817-829 freezes everything but `mm_projector`; the loss is HF's causal-LM cross-entropy of
 * ref:vis_zephyr/model/language_model/vis_zephyr.py:51-98 (shift by one, ignore_index -100, mean over the valid targets);
 * ref:vis_zephyr/train/vis_zephyr_trainer.py:224-302 + ref:script/pretrain.sh:39-42: AdamW on the projector parameters,
 * DeepSpeed ZeRO-2 data parallelism.  A trainer belongs to one finalized tp_size 1 engine and owns: W^T copies of the frozen
 * Zephyr linears, fp32 master / Adam moments / gradient arenas of the 165 projector tensors (engine names "qf.*", engine
 * layout - vz_hip/train.py maps them to the reference's parameter names), and an arena of saved activations + scratch.
 *   vz_train_stage1_accumulate  forward (activations kept) + backward of ONE micro-batch; gradients ACCUMULATE.  Stage inputs
 *       as the inference entry points take them (vz_clip_fused_features / vz_qformer / vz_embed_splice / vz_llm_prefill) plus
 *       d_vis_rows int32 [T*32] (row of [B*S] each visual token was spliced into, -1 = cut off), d_labels int32 [B,S] (HF
 *       labels; the shift happens inside) and inv_n = 1 / (valid targets of the whole batch).
 *   vz_train_loss_sum           sum over the rows of the last micro-batch of (logsumexp - target logit)  (blocking)
 *   vz_train_allreduce          RCCL all-reduce (ncclAvg: the mean over the data-parallel ranks of their mean-loss gradients, as
 *                               DeepSpeed / HF Trainer reduce them) of the flat gradient arena, 256 MiB buckets; no-op without vz_train_comm_init
 *   vz_train_adamw_step         torch.optim.AdamW (no amsgrad) on every projector tensor; rewrites the engine's working copies,
 *                               clears the gradients
 * ------------------------------------------------------------------------------------------ */
typedef struct vz_trainer vz_trainer;
int vz_train_create(vz_engine* e, vz_trainer** out, vz_stream stream);
int vz_train_destroy(vz_trainer* t);
int vz_train_set_master(vz_trainer* t, const char* engine_name, const float* d_values, long n, vz_stream stream);
int vz_train_param_count(vz_trainer* t);
int vz_train_param_info(vz_trainer* t, int i, const char** name, long* n, long* offset, int* is_matrix);
int vz_train_arenas(vz_trainer* t, float** d_grad, float** d_master, float** d_m, float** d_v, long* total_floats);
int vz_train_stage1_accumulate(vz_trainer* t, const void* d_images, int T, const void* d_text, int n_samples, int Lmax,
                               const int* h_tile_sample, const int* d_kind, const int* d_idx, const int* d_vis_rows, int B, int S,
                               const int* h_seqlens, const int* d_pos, const int* d_labels, float inv_n, vz_stream stream);
int vz_train_loss_sum(vz_trainer* t, double* out, vz_stream stream);
int vz_train_zero_grad(vz_trainer* t, vz_stream stream);
int vz_train_comm_init(vz_trainer* t, const char* id128, int rank, int world);
int vz_train_allreduce(vz_trainer* t, vz_stream stream);
int vz_train_adamw_step(vz_trainer* t, float lr, float beta1, float beta2, float eps, float weight_decay, vz_stream stream);

/* ViP "point" overlay on the device (ref:vis_zephyr/model/vip_processor/conversation_generator.py:143-153,170-175: the
 * `vcr_qa` / `vcr_qar` visual prompt = `ImageDraw.ellipse(box, fill=rgba, outline=rgba)` on a transparent canvas +
 * `Image.alpha_composite` + convert("RGB")): composites one filled ellipse with the INTEGER box (x0, y0, x1, y1) - the
 * reference's float box truncated as Pillow's (int) cast does - and colour rgba (r | g << 8 | b << 16 | a << 24) onto the
 * u8 [h, w, 3] image in place, bit-exact with Pillow (oracle/vip_oracle.py).  Points of one image are applied in call order. */
int vz_op_vip_point(void* d_image_u8, int h, int w, int x0, int y0, int x1, int y1, unsigned rgba, vz_stream stream);

/* argmax over fp32 logits rows: ids int32 [rows] (first maximal index) */
int vz_op_argmax(const float* d_logits, int rows, int cols, int* d_ids, vz_stream stream);

/* A/B hook of the bench tools (process-wide; production values in brackets).  Knobs:
 *   0  GEMV variant [0 = production choice; 1.. = alternatives compiled in: rows per wave, chunks in flight, non-temporal loads]
 *   1  tile GEMM choice [0 = by grid size; 1 = always 128^2; 2 = always 256^2]      2  prefill attention generation [3]
 *   3  split-K of M <= 512 linears [0 = auto; 1 = never]                             4  256^2 GEMM stream-K tail [1; 0 = whole tiles only; 2 = forced]
 *   5  stream-K skew in K-tiles                                                     6  record 256^2 GEMM phase stamps (vz_prof_gemm_stamps)
 *   7  route the collectives of a tp_size == 1 engine holding a one-rank communicator through RCCL (self-test; 2 = shape rehearsal)
 *   9  2..16-row linears [1 = MFMA weight stream; 0 = GEMV / tile GEMM; 5 = ignore the tiled weight copies]
 *  10  context splits of the fused decode attention [0 = engine default; 1..64]     11  256^2 GEMM workgroups wait for their stores (experiment) [0]
 *  14  rows from which a decode step's linears take the 128^2 tile GEMM [29]          15  split-K factor of the K = 4096 projections on that route [8]
 *  16  record stage stamps of the prefill attention kernel (vz_prof_attn_stamps)    19  gemm_wide.hip for 17..64-row gate|up / lm_head [1]
 *  21  fp8 tile GEMM choice [0 = by grid size; 1 = 128^2; 2 = 256^2]                22  rows from which vz_engine_prefill_fp8 engines take the fp8 MFMA [768]
 *  23  key split of few-row head_dim-512 attention [0 = every 96 keys; 1 = never; n >= 2 = n splits]
 *  24  most K slices of an M <= 512 linear [8]                                      25  Q-Former cross-attention K|V of all blocks as one GEMM [1]
 *  26  K slices for tile-GEMM grids that leave a CU one workgroup (M > 512, < 256 tiles) [1; 0 = whole-K tiles: batch-invariant]
 *  27  K splits of the e4m3 17..64-row stream [0 = by shape]                        28  batch-1 decode steps as ONE resident grid per token (decode_persist.hip) [0]
 *  29  one-shot all-reduce for the TP decode step when peer areas are attached [1]   30  batch-1 decode attention + O projection as one launch (attn_o_fused.hip) [1]
 *  31  that launch's O role: wait (x ~0.2 us) before it requests its weights [12]    32  training step: tile-resident attention backward for head_dim 128 [1]
 *  33  prefill: RoPE of the queries inside the attention's Q load [1; 0 = a rotated copy of Q from rope_kv_kernel; bit-identical]
 *  34  256^2 GEMM: persistent whole-tile workgroups when a launch has more whole tiles than CUs [1; 0 = one workgroup per tile; bit-identical]
 *  35  3..16-row persistent weight stream: the same number of row groups on every workgroup [1; 0 = one workgroup per CU; bit-identical] */
int vz_tune_set(int knob, int value);

/* per-kernel-class timing of the engine's launches with HIP events on the launch stream (bench.py's roofline
 * leg).  enable=1 disables graph replay and brackets every launch of class `klass` with events.
 * classes: 0 gemm(mfma) 1 gemv 2 attention 3 attn_decode 4 norm 5 other 6 fused attention half 7 RCCL collectives */
int vz_prof_enable(vz_engine* e, int enable, int klass);
/* synchronises the events; returns launches counted and their total milliseconds */
int vz_prof_read(vz_engine* e, long* n_launches, double* total_ms);
/* in-kernel phase stamps (s_memrealtime, 100 MHz) of the last 256^2 GEMM launched with vz_tune_set(6, 1):
 * 16 int64 per workgroup = {start, then per K-slice: loop begin, loop end, fix-up end, epilogue end, (nk<<32 | flags)} */
int vz_prof_gemm_stamps(long long* host_out, int max_wgs, int* n_wgs);
/* stage cycle counts (s_memtime) of one wave of the prefill attention kernel, last launch with vz_tune_set(16, 1): the longest
 * causal workgroup's wave 0 - [issue K/V loads, QK^T, softmax, PV, wait, barrier, -, loop top, tiles] (16 int64), followed by 2048 x 4 int64
 * of per-workgroup schedule: start, end (s_memrealtime, 100 MHz), XCC id << 32 | HW_ID, query block << 32 | tiles.  host buffer: 16 + 8192 int64 */
int vz_prof_attn_stamps(long long* host16);

#ifdef __cplusplus
}
#endif
#endif /* VISZEPHYR_H */
