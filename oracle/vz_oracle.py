"""CPU oracle for the Vision-Zephyr forward/generate hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  The shipped path (vision-zephyr_amd/) never does
and fails loudly when its HIP library is missing.

It is a from-scratch functional restatement (plain tensors + a dict of reference-named weights,
torch CPU fp32) of what the reference computes on this path:

  a8/a9  CLIP ViT-L/14-336 tower, all hidden states   hf:models/clip/modeling_clip.py:138-219,259-385,594-658
         (called from ref:vis_zephyr/model/vision_encoder/vision_encoder.py:80-117)
  a10    multi-layer fusion                            ref:vis_zephyr/model/vision_encoder/vision_encoder.py:58-78
                                                       ref:vis_zephyr/model/gating_fusion/gating_fusion.py:22-50
  a11    Q-Former projector                            ref:vis_zephyr/model/multimodal_projector/builder.py:12-92
         (torch.nn.MultiheadAttention, batch_first, no masks)
  a5     encode_images                                 ref:vis_zephyr/model/vis_zephyr_arch.py:120-124
  a6/a7  prepare_inputs_labels_for_multimodal (splice) ref:vis_zephyr/model/vis_zephyr_arch.py:129-333,396-530
  a12    Mistral decoder forward with KV cache         hf:models/mistral/modeling_mistral.py:35-466
  a3/a13 greedy generate (new tokens only)             ref:vis_zephyr/model/language_model/vis_zephyr.py:100-142
                                                       hf:generation/utils.py (_sample, greedy branch)

PINNING: `oracle/pin_against_reference.py` imports the reference from /root/reference in the
build container, loads the same hash-generated weights into it and checks every stage of this
file against it in fp32 (<= 1e-5 relative); the resulting vectors are committed under
tests/golden/.  The reference itself has no tests or golden vectors for this path (SURVEY.md
section 4), so those fixtures are what pins the oracle.

Three precisions:
  * FP32 = Prec(bf16=False): fp32 everywhere - the reference's CPU float32 path (BASELINE.json configs[0]).
  * W16  = Prec(False, weights=True): fp32 arithmetic on bf16-rounded matrices / embeddings / input tiles - the model a
    bf16 checkpoint holds.  FP32 -> W16 is weight rounding (not the implementation's error); W16 -> HIP is the
    implementation's own (activation) rounding.
  * BF16 = Prec(bf16=True):  same arithmetic in fp32, but every tensor the HIP path writes to HBM as bf16
    is rounded to bf16 at that point (matrix weights, GEMM/attention/norm outputs): the bf16 band the GPU
    path is expected to sit in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100          # ref:vis_zephyr/constants.py:11
IMAGE_TOKEN_INDEX = -200     # ref:vis_zephyr/constants.py:13

VT = "model.vision_tower.vision_tower.vision_model."
QF = "model.mm_projector."


class Prec:
    """Rounding policy.  `acts`: round every tensor the HIP path writes to HBM as bf16 (GEMM / attention / norm outputs,
    inputs); `weights`: round the matrices / embeddings the HIP path stores as bf16 (vectors - bias, norm scale - stay fp32).
    Three policies are used: FP32 (neither: the reference's CPU float32 path), W16 (weights only: fp32 arithmetic on the bf16
    model a checkpoint actually holds - separates weight rounding from the implementation's own activation rounding) and BF16
    (both: the bf16 band the HIP path is expected to sit in)."""

    def __init__(self, bf16: bool = False, weights: Optional[bool] = None, act_fp8: bool = False):
        self.bf16 = bf16
        self.bf16_weights = bf16 if weights is None else weights
        # fp8 MFMA prefill of the W8A16 engine (gemm_fp8.hip): the INPUT of every Zephyr prefill linear is rounded to e4m3 with one
        # power-of-two scale per row (the weights' quantiser, fake_quantize_rows below); decode steps keep bf16 activations
        self.act_fp8 = act_fp8

    def q_in(self, x: torch.Tensor, prefill: bool) -> torch.Tensor:
        if not (self.act_fp8 and prefill):
            return x
        return fake_quantize_rows(x.reshape(-1, x.shape[-1])).reshape(x.shape)

    def r(self, x: torch.Tensor) -> torch.Tensor:
        if not self.bf16:
            return x
        return x.to(torch.bfloat16).to(torch.float32)

    def w(self, x: torch.Tensor) -> torch.Tensor:
        if not self.bf16_weights:
            return x
        return x.to(torch.bfloat16).to(torch.float32)

    def x_in(self, x: torch.Tensor) -> torch.Tensor:
        """model inputs (image tiles): the HIP boundary takes bf16 tiles, so W16 rounds them too."""
        if not (self.bf16 or self.bf16_weights):
            return x
        return x.to(torch.bfloat16).to(torch.float32)


FP32 = Prec(False)
BF16 = Prec(True)
W16 = Prec(False, weights=True)
FP32_FP8ACT = Prec(False, act_fp8=True)      # fp32 arithmetic on e4m3-quantised prefill activations (use with quantize_state_dict)
BF16_FP8ACT = Prec(True, act_fp8=True)       # + bf16 rounding at the HIP path's store points


def _lin(x, w, b, P: Prec):
    """x @ w^T (+ b) with fp32 accumulation; caller rounds the (possibly fused) result."""
    y = x @ P.w(w).t()
    if b is not None:
        y = y + b
    return y


def _layernorm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ================================================================================================
# a8/a9: CLIP vision tower  (hf:models/clip/modeling_clip.py)
# ================================================================================================
def clip_hidden_states(cfg, sd: Dict[str, torch.Tensor], images: torch.Tensor, P: Prec = FP32
                       ) -> List[torch.Tensor]:
    """images [T,3,336,336] -> list of clip_layers+1 tensors [T,577,C].
    [0] = pre_layrnorm(embeddings), [i] = output of encoder layer i;
    post_layernorm is never applied on this path."""
    T = images.shape[0]
    C, p = cfg.clip_hidden, cfg.clip_patch
    if images.shape[-1] != cfg.clip_image or images.shape[-2] != cfg.clip_image:
        raise ValueError(f"Input image size ({images.shape[-2]}*{images.shape[-1]}) doesn't match model "
                         f"({cfg.clip_image}*{cfg.clip_image}).")   # hf:...modeling_clip.py:204-207
    x = P.x_in(images.to(torch.float32))
    g = cfg.clip_image // p
    # Conv2d(3->C, k=p, s=p, no bias) == GEMM over im2col patches (row-major over the grid)
    patches = x.view(T, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(T, g * g, 3 * p * p)
    wpe = sd[VT + "embeddings.patch_embedding.weight"].reshape(C, 3 * p * p)
    pe = P.r(_lin(patches, wpe, None, P))                                   # [T,576,C]
    cls = P.w(sd[VT + "embeddings.class_embedding"]).view(1, 1, C).expand(T, 1, C)
    pos = P.w(sd[VT + "embeddings.position_embedding.weight"]).unsqueeze(0)
    h = P.r(torch.cat([cls, pe], dim=1) + pos)                              # [T,577,C]
    h = P.r(_layernorm(h, sd[VT + "pre_layrnorm.weight"], sd[VT + "pre_layrnorm.bias"], cfg.clip_eps))
    hs = [h]
    nh = cfg.clip_heads
    hd = C // nh
    for i in range(cfg.clip_layers):
        pre = VT + f"encoder.layers.{i}."
        y = P.r(_layernorm(h, sd[pre + "layer_norm1.weight"], sd[pre + "layer_norm1.bias"], cfg.clip_eps))
        q = P.r(_lin(y, sd[pre + "self_attn.q_proj.weight"], sd[pre + "self_attn.q_proj.bias"], P))
        k = P.r(_lin(y, sd[pre + "self_attn.k_proj.weight"], sd[pre + "self_attn.k_proj.bias"], P))
        v = P.r(_lin(y, sd[pre + "self_attn.v_proj.weight"], sd[pre + "self_attn.v_proj.bias"], P))
        a = _attention(q.view(T, -1, nh, hd), k.view(T, -1, nh, hd), v.view(T, -1, nh, hd),
                       scale=hd ** -0.5, P=P).reshape(T, -1, C)
        h = P.r(_lin(a, sd[pre + "self_attn.out_proj.weight"], sd[pre + "self_attn.out_proj.bias"], P) + h)
        y = P.r(_layernorm(h, sd[pre + "layer_norm2.weight"], sd[pre + "layer_norm2.bias"], cfg.clip_eps))
        f = _lin(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"], P)
        f = P.r(f * torch.sigmoid(1.702 * f))                               # quick_gelu, hf:activations.py:117-123
        h = P.r(_lin(f, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"], P) + h)
        hs.append(h)
    return hs


def _attention(q, k, v, scale, P: Prec, mask: Optional[torch.Tensor] = None):
    """q [B,Sq,H,D], k/v [B,Sk,Hkv,D] -> [B,Sq,H,D]; softmax in fp32; GQA by head grouping
    (hf:models/mistral/modeling_mistral.py:84-119).  `mask` broadcastable bool [B,1,Sq,Sk], True = keep.
    bf16 policy: probabilities are rounded to bf16 before P@V (the HIP kernels feed P to the MFMA
    as bf16), the normaliser is accumulated in fp32 from the unrounded exponentials."""
    B, Sq, H, D = q.shape
    Hkv = k.shape[2]
    grp = H // Hkv
    qh = q.permute(0, 2, 1, 3)                                  # [B,H,Sq,D]
    kh = k.permute(0, 2, 1, 3).repeat_interleave(grp, dim=1)
    vh = v.permute(0, 2, 1, 3).repeat_interleave(grp, dim=1)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    m = s.amax(dim=-1, keepdim=True)
    e = torch.exp(s - m)
    l = e.sum(dim=-1, keepdim=True)
    o = (P.r(e) @ vh) / l
    return P.r(o.permute(0, 2, 1, 3))


# ================================================================================================
# a10: fusion  (ref:vis_zephyr/model/vision_encoder/vision_encoder.py:58-78,
#               ref:vis_zephyr/model/gating_fusion/gating_fusion.py:22-50)
# ================================================================================================
def fusion(cfg, hidden_states: Sequence[torch.Tensor], P: Prec = FP32, select_feature: str = "patch"):
    n_sel = cfg.fusion_groups * cfg.fusion_layers_per_group + 1
    sel = list(hidden_states[-n_sel:])
    if select_feature == "patch":
        sel = [t[:, 1:] for t in sel]
    elif select_feature != "cls_patch":
        raise ValueError(f"Unknown feature selection strategy: {select_feature}")
    G = cfg.fusion_groups
    if len(sel) < G + 1:
        raise ValueError(f"Expected at least {G + 1} feature tensors, got {len(sel)}.")
    last, inter = sel[-1], sel[:-1]
    if len(inter) % G != 0:
        raise ValueError(f"Number of intermediate features ({len(inter)}) must be divisible by num_groups ({G}).")
    per = len(inter) // G
    groups = [torch.stack(inter[i * per:(i + 1) * per], 0).mean(0) for i in range(G)]
    return P.r(torch.cat(groups + [last], dim=-1))


def clip_tower(cfg, sd, images, P: Prec = FP32, select_feature: Optional[str] = None):
    """a8: CLIPVisionTower.forward for a 4-D batch -> [T,576,5*C] ('patch') or [T,577,5*C] ('cls_patch',
    ref:vis_zephyr/model/vision_encoder/vision_encoder.py:66-73); default follows cfg.clip_keep_cls."""
    if select_feature is None:
        select_feature = "cls_patch" if getattr(cfg, "clip_keep_cls", False) else "patch"
    return fusion(cfg, clip_hidden_states(cfg, sd, images, P), P, select_feature)


# ================================================================================================
# a11: Q-Former  (ref:vis_zephyr/model/multimodal_projector/builder.py:12-92)
# ================================================================================================
def _mha(xq, xkv, wq, wk, wv, bq, bk, bv, wo, bo, nheads, P: Prec):
    """torch.nn.MultiheadAttention forward (batch_first, no masks, eval)."""
    B, Nq, E = xq.shape
    hd = E // nheads
    q = P.r(_lin(xq, wq, bq, P)).view(B, Nq, nheads, hd)
    k = P.r(_lin(xkv, wk, bk, P)).view(B, -1, nheads, hd)
    v = P.r(_lin(xkv, wv, bv, P)).view(B, -1, nheads, hd)
    a = _attention(q, k, v, scale=hd ** -0.5, P=P).reshape(B, Nq, E)
    return _lin(a, wo, bo, P)                                      # caller adds the residual and rounds


def qformer_block(cfg, sd, i: int, x, feats, P: Prec):
    H = cfg.hidden
    p = QF + f"blocks.{i}."
    y = P.r(_layernorm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.qf_eps))
    w, b = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
    x = P.r(x + _mha(y, y, w[:H], w[H:2 * H], w[2 * H:], b[:H], b[H:2 * H], b[2 * H:],
                     sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"], cfg.qf_heads, P))
    y = P.r(_layernorm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.qf_eps))
    b = sd[p + "cross_attn.in_proj_bias"]
    x = P.r(x + _mha(y, feats, sd[p + "cross_attn.q_proj_weight"], sd[p + "cross_attn.k_proj_weight"],
                     sd[p + "cross_attn.v_proj_weight"], b[:H], b[H:2 * H], b[2 * H:],
                     sd[p + "cross_attn.out_proj.weight"], sd[p + "cross_attn.out_proj.bias"], cfg.qf_heads, P))
    y = P.r(_layernorm(x, sd[p + "norm3.weight"], sd[p + "norm3.bias"], cfg.qf_eps))
    f = _lin(y, sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"], P)
    f = P.r(F.gelu(f))                                              # exact erf GELU (nn.GELU default)
    x = P.r(x + _lin(f, sd[p + "ffn.2.weight"], sd[p + "ffn.2.bias"], P))
    return x


def qformer(cfg, sd, feats, text_embeddings: Optional[torch.Tensor], P: Prec = FP32,
            return_blocks: bool = False):
    """feats [T,576,5120], text_embeddings [T,Lmax,H] or None -> [T,32,H].
    Block 0 runs on [queries ; text] with no key-padding mask."""
    T = feats.shape[0]
    f = P.r(_layernorm(feats, sd[QF + "pre_norm.weight"], sd[QF + "pre_norm.bias"], cfg.qf_eps))
    q = P.w(sd[QF + "learned_queries"]).unsqueeze(0).expand(T, -1, -1)
    x = torch.cat([q, P.r(text_embeddings)], dim=1) if text_embeddings is not None else q
    x = qformer_block(cfg, sd, 0, x, f, P)[:, :cfg.qf_queries]
    per_block = [x]
    for i in range(1, cfg.qf_blocks):
        x = qformer_block(cfg, sd, i, x, f, P)
        per_block.append(x)
    out = P.r(_layernorm(x, sd[QF + "norm.weight"], sd[QF + "norm.bias"], cfg.qf_eps))
    return (out, per_block) if return_blocks else out


def encode_images(cfg, sd, images, text_embeddings, P: Prec = FP32):
    """a5: ref:vis_zephyr/model/vis_zephyr_arch.py:120-124."""
    return qformer(cfg, sd, clip_tower(cfg, sd, images, P), text_embeddings, P)


# ================================================================================================
# W8A16 weights (SURVEY config 5): CPU restatement of vz_hip/quant.py, applied to the reference-named state dict
# ================================================================================================
_FP8_KEYS = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
             "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")


def fake_quantize_rows(w: torch.Tensor) -> torch.Tensor:
    """one power-of-two scale per output row, weights rounded to OCP e4m3 (RNE): w -> 2^e * e4m3(w * 2^-e), e = ceil(log2(amax/448))."""
    w = w.float()
    amax = w.abs().amax(dim=1)
    e = torch.where(amax > 0, torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / 448.0)), torch.zeros_like(amax))
    q = (w * torch.exp2(-e).unsqueeze(1)).to(torch.float8_e4m3fn).float()
    return q * torch.exp2(e).unsqueeze(1)


def quantize_state_dict(sd):
    """the model the weight_fp8 engine computes with: every Zephyr linear + lm_head row-quantised (rows are independent, so
    quantising q/k/v or gate/up before the engine stacks them gives the same bytes); embeddings, norms, CLIP, Q-Former untouched.
    The quantiser starts from the bf16 weights (what a checkpoint holds and what the engine stores), not from fp32 masters."""
    out = dict(sd)
    for k, v in sd.items():
        if k == "lm_head.weight" or (k.startswith("model.layers.") and k.endswith(_FP8_KEYS)):
            out[k] = fake_quantize_rows(v.bfloat16().float())
    return out


# NF4 (`load_4bit`): CPU restatement of bitsandbytes' published nf4 quantiser (QLoRA appendix E; blocksize 64, nearest level, fp32 absmax -
# the double quantisation of the absmax values is NOT restated; bitsandbytes is not in the reference tree nor installed: parity unpinned)
_NF4 = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
        -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
        0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)


def fake_quantize_nf4(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> nearest NF4 level * absmax per block of 64 consecutive row elements (fp32); plain loops over the levels, no bucketize"""
    w = w.float()
    N, K = w.shape
    x = w.reshape(N, K // 64, 64)
    absmax = x.abs().amax(dim=2, keepdim=True)
    xn = x / torch.clamp(absmax, min=1e-30)
    best = torch.full_like(xn, _NF4[0])
    err = (xn - _NF4[0]).abs()
    for lv in _NF4[1:]:                                   # ascending levels, strict improvement: a tie keeps the LOWER level
        e = (xn - lv).abs()
        take = e < err
        best = torch.where(take, torch.full_like(xn, lv), best)
        err = torch.where(take, e, err)
    return (best * absmax).reshape(N, K)


def quantize_state_dict_nf4(sd):
    """the model a weight_nf4 engine computes with: the decoder layers' seven linears NF4-quantised from their bf16 values and rounded
    back to bf16; lm_head, embeddings, norms, CLIP and the Q-Former untouched (bitsandbytes' default skip list keeps lm_head)."""
    out = dict(sd)
    for k, v in sd.items():
        if k.startswith("model.layers.") and k.endswith(_FP8_KEYS):
            out[k] = fake_quantize_nf4(v.bfloat16().float()).bfloat16().float()
    return out


# ================================================================================================
# a6/a7: embedding splice  (ref:vis_zephyr/model/vis_zephyr_arch.py:129-333,396-530)
# ================================================================================================
def embed_tokens(sd, ids: torch.Tensor, P: Prec):
    return P.w(sd["model.embed_tokens.weight"])[ids]


def prepare_inputs_labels_for_multimodal(cfg, sd, input_ids, position_ids, attention_mask, past_key_values,
                                         labels, images, images_size=None, P: Prec = FP32,
                                         merge_type: str = "flat", max_length: Optional[int] = None,
                                         padding_side: str = "right", encode_fn=None):
    """Returns (None, position_ids|None, attention_mask|None, past_key_values, inputs_embeds, labels|None).
    `images`: list of [N_i,3,336,336] (or [3,336,336]) tensors, or a 5-D tensor [B,N,3,336,336].
    `encode_fn(images[T,...], text_emb[T,Lmax,H]) -> [T,32,H]` overrides the oracle's own encoder
    (tests use it to splice HIP-computed features through the oracle's index logic)."""
    if images is None or input_ids.shape[1] == 1:                        # :148-149
        return input_ids, position_ids, attention_mask, past_key_values, None, labels
    if not (isinstance(images, (list, tuple)) or images.ndim == 5):
        raise NotImplementedError("4-D `images` is a dead path in the reference (SURVEY.md Appendix A Q7)")
    if isinstance(images, (list, tuple)):
        images = [x.unsqueeze(0) if x.ndim == 3 else x for x in images]
    # --- stage 1: per-sample text embeddings (pads included, Q4), zero-padded to Lmax (Q3), per tile ---
    tes = []
    for i, tiles in enumerate(images):
        ids_i = input_ids[i]
        te = embed_tokens(sd, ids_i[ids_i != IMAGE_TOKEN_INDEX], P)      # [L_i,H]
        tes.append(te.unsqueeze(0).expand(tiles.shape[0], -1, -1))
    Lmax = max(t.shape[1] for t in tes)
    tes = [torch.cat([t, t.new_zeros(t.shape[0], Lmax - t.shape[1], t.shape[2])], 1) if t.shape[1] < Lmax else t
           for t in tes]
    text_emb = torch.cat(tes, 0)
    cat_images = torch.cat(list(images), 0)
    feats = (encode_fn or (lambda im, te: encode_images(cfg, sd, im, te, P)))(cat_images, text_emb)
    per_sample = list(torch.split(feats, [t.shape[0] for t in images], 0))
    if merge_type == "flat":                                             # :407-413
        per_sample = [f.flatten(0, 1) for f in per_sample]
    elif merge_type.startswith("spatial"):
        raise NotImplementedError("spatial merge types are unreachable with the Q-Former (Appendix A Q5)")
    else:
        raise ValueError(f"Unknown mm_patch_merge_type: {merge_type}")
    # --- stage 2: splice ---
    _labels, _pos, _mask = labels, position_ids, attention_mask
    mask = torch.ones_like(input_ids, dtype=torch.bool) if attention_mask is None else attention_mask.bool()
    if labels is None:
        labels = torch.full_like(input_ids, IGNORE_INDEX)
    B = input_ids.shape[0]
    new_emb, new_lab = [], []
    img_i = 0
    for b in range(B):
        ids = input_ids[b][mask[b]]
        lab = labels[b][mask[b]]
        n_img = int((ids == IMAGE_TOKEN_INDEX).sum())
        if n_img == 0:                                                    # Q11: still consumes a feature slot
            new_emb.append(embed_tokens(sd, ids, P))
            new_lab.append(lab)
            img_i += 1
            continue
        cuts = [-1] + torch.where(ids == IMAGE_TOKEN_INDEX)[0].tolist() + [ids.shape[0]]
        pe, pl = [], []
        for j in range(len(cuts) - 1):
            seg = ids[cuts[j] + 1:cuts[j + 1]]
            pe.append(embed_tokens(sd, seg, P))
            pl.append(lab[cuts[j] + 1:cuts[j + 1]])
            if j < n_img:
                f = per_sample[img_i]
                img_i += 1
                pe.append(f)
                pl.append(torch.full((f.shape[0],), IGNORE_INDEX, dtype=lab.dtype))
        new_emb.append(torch.cat(pe, 0))
        new_lab.append(torch.cat(pl, 0))
    if max_length is not None:                                            # :308-313
        new_emb = [x[:max_length] for x in new_emb]
        new_lab = [x[:max_length] for x in new_lab]
    Smax = max(x.shape[0] for x in new_emb)
    emb = torch.zeros(B, Smax, cfg.hidden)
    lab_out = torch.full((B, Smax), IGNORE_INDEX, dtype=new_lab[0].dtype)
    m_out = torch.zeros(B, Smax, dtype=torch.bool)
    pos_out = torch.zeros(B, Smax, dtype=torch.long)
    for b in range(B):
        n = new_emb[b].shape[0]
        sl = slice(Smax - n, Smax) if padding_side == "left" else slice(0, n)
        if n > 0:
            emb[b, sl] = new_emb[b]
            lab_out[b, sl] = new_lab[b]
            m_out[b, sl] = True
            pos_out[b, sl] = torch.arange(n)
    if _mask is not None:
        m_out = m_out.to(_mask.dtype)
    return (None, pos_out if _pos is not None else None, m_out if _mask is not None else None,
            past_key_values, emb, lab_out if _labels is not None else None)


# ================================================================================================
# a12: Mistral decoder  (hf:models/mistral/modeling_mistral.py:35-466)
# ================================================================================================
def rmsnorm(x, w, eps):
    v = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(v + eps))


def rope_tables(cfg, positions: torch.Tensor):
    """positions [..] int64 -> cos,sin [..,head_dim] fp32 (rotate-half convention, :51-81,262-317)."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    fr = positions.to(torch.float32).unsqueeze(-1) * inv
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


@dataclass
class KVCache:
    k: List[torch.Tensor]     # per layer [B,ctx,Hkv,D]
    v: List[torch.Tensor]
    mask: torch.Tensor        # [B,ctx] bool: which cached positions are real tokens


def llm_forward(cfg, sd, inputs_embeds, attention_mask=None, position_ids=None, cache: Optional[KVCache] = None,
                P: Prec = FP32, last_only: bool = False, return_hidden: bool = False):
    """inputs_embeds [B,S,H] -> (logits [B,S,V] fp32, cache).  `attention_mask` [B,ctx_new] (ctx_new = cached + S)
    marks real tokens; causal + sliding-window mask as HF builds it.  With `last_only`, lm_head runs on
    the last position only (what generate needs)."""
    B, S, H = inputs_embeds.shape
    past = 0 if cache is None else cache.k[0].shape[1]
    ctx = past + S
    if attention_mask is None:
        attention_mask = torch.ones(B, ctx, dtype=torch.bool)
    attention_mask = attention_mask.bool()
    if position_ids is None:
        position_ids = torch.arange(past, ctx).unsqueeze(0).expand(B, S)
    cos, sin = rope_tables(cfg, position_ids)                 # [B,S,D]
    cos, sin = cos.unsqueeze(2), sin.unsqueeze(2)
    qpos = torch.arange(past, ctx).view(1, 1, S, 1)
    kpos = torch.arange(ctx).view(1, 1, 1, ctx)
    keep = (kpos <= qpos) & (kpos > qpos - cfg.sliding_window) & attention_mask.view(B, 1, 1, ctx)
    # a fully masked (padding) query row would softmax over nothing; HF leaves it attending uniformly
    dead = ~keep.any(-1, keepdim=True)
    keep = keep | dead
    nh, nkv, hd = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
    x = P.r(inputs_embeds.to(torch.float32))
    newk, newv = [], []
    for i in range(cfg.n_layers):
        p = f"model.layers.{i}."
        y = P.q_in(P.r(rmsnorm(x, sd[p + "input_layernorm.weight"], cfg.rms_eps)), cache is None)
        q = P.r(_lin(y, sd[p + "self_attn.q_proj.weight"], None, P)).view(B, S, nh, hd)
        k = P.r(_lin(y, sd[p + "self_attn.k_proj.weight"], None, P)).view(B, S, nkv, hd)
        v = P.r(_lin(y, sd[p + "self_attn.v_proj.weight"], None, P)).view(B, S, nkv, hd)
        q = P.r(q * cos + _rot_half(q) * sin)
        k = P.r(k * cos + _rot_half(k) * sin)
        if cache is not None:
            k = torch.cat([cache.k[i], k], 1)
            v = torch.cat([cache.v[i], v], 1)
        newk.append(k)
        newv.append(v)
        a = P.q_in(_attention(q, k, v, scale=hd ** -0.5, P=P, mask=keep).reshape(B, S, nh * hd), cache is None)
        x = P.r(_lin(a, sd[p + "self_attn.o_proj.weight"], None, P) + x)
        y = P.q_in(P.r(rmsnorm(x, sd[p + "post_attention_layernorm.weight"], cfg.rms_eps)), cache is None)
        g = _lin(y, sd[p + "mlp.gate_proj.weight"], None, P)
        u = _lin(y, sd[p + "mlp.up_proj.weight"], None, P)
        a = P.q_in(P.r(F.silu(g) * u), cache is None)
        x = P.r(_lin(a, sd[p + "mlp.down_proj.weight"], None, P) + x)
    hfin = P.r(rmsnorm(x, sd["model.norm.weight"], cfg.rms_eps))
    hl = hfin[:, -1:] if last_only else hfin
    logits = _lin(hl, sd["lm_head.weight"], None, P)
    new_cache = KVCache(newk, newv, attention_mask)
    if return_hidden:
        return logits, new_cache, hfin
    return logits, new_cache


def greedy_generate(cfg, sd, inputs_embeds, max_new_tokens: int, eos_token_id=None, P: Prec = FP32,
                    return_logits: bool = False):
    """a3/a13 for batch 1.. greedy: prefill on inputs_embeds, then token-by-token with the cache.
    Returns new tokens only [B,n_new] (HF semantics with inputs_embeds).
    `eos_token_id` (int | list | None): stop when every row has produced one (rows that finished
    earlier keep emitting eos as pad, as HF does with pad_token_id = eos)."""
    B = inputs_embeds.shape[0]
    eos = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    logits, cache = llm_forward(cfg, sd, inputs_embeds, P=P, last_only=True)
    out, all_logits = [], []
    done = torch.zeros(B, dtype=torch.bool)
    pad = min(eos) if eos else 0
    for step in range(max_new_tokens):
        last = logits[:, -1].float()
        all_logits.append(last)
        nxt = last.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
        out.append(nxt)
        if eos:
            done = done | torch.tensor([int(t) in eos for t in nxt])
            if bool(done.all()):
                break
        if step + 1 == max_new_tokens:
            break
        emb = embed_tokens(sd, nxt, P).unsqueeze(1)
        mask = torch.cat([cache.mask, torch.ones(B, 1, dtype=torch.bool)], 1)
        logits, cache = llm_forward(cfg, sd, emb, attention_mask=mask, cache=cache, P=P, last_only=True)
    ids = torch.stack(out, 1)
    return (ids, torch.stack(all_logits, 1)) if return_logits else ids


def generate(cfg, sd, input_ids, images, max_new_tokens, eos_token_id=None, P: Prec = FP32, **kw):
    """ref:vis_zephyr/model/language_model/vis_zephyr.py:100-142 (greedy)."""
    if images is not None:
        _, _, _, _, emb, _ = prepare_inputs_labels_for_multimodal(cfg, sd, input_ids, None, None, None, None,
                                                                   images, P=P)
    else:
        emb = embed_tokens(sd, input_ids, P)
    return greedy_generate(cfg, sd, emb, max_new_tokens, eos_token_id, P, **kw)
