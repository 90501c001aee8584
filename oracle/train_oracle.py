"""CPU oracle of the Stage-1 pretrain step - TEST INFRASTRUCTURE, not product code.

Stage 1 trains the Q-Former projector only (`ref:vis_zephyr/train/train.py:817-829`: everything frozen, then
`model.get_model().mm_projector.parameters()` re-enabled), on the token cross-entropy of the caption through the frozen Zephyr
(`ref:vis_zephyr/model/language_model/vis_zephyr.py:51-98` hands `labels` to HF's `MistralForCausalLM.forward`, whose
`ForCausalLMLoss` shifts by one, ignores -100 and takes the mean over the remaining positions), with HF Trainer's AdamW
(`ref:script/pretrain.sh:16,38-41`: the projector groups run at --mm_projector_lr 2e-3 - `ref:vis_zephyr/train/vis_zephyr_trainer.py:224-302`;
the 2e-5 of --learning_rate only reaches groups that are empty in Stage 1 - weight decay 0, cosine schedule with 3 % warm-up; betas
0.9 / 0.999, eps 1e-8).

The forward is the pinned restatement in vz_oracle.py; the gradients are torch autograd THROUGH that restatement (the CLIP tower
runs under no_grad exactly as `ref:vis_zephyr/model/vision_encoder/vision_encoder.py:80` decorates it), so this file adds no
arithmetic of its own beyond the loss and the optimiser update.  Pinned against the reference's own `loss.backward()` and
`torch.optim.AdamW` by oracle/pin_train_step.py (tests/golden/stage1_step.npz).  The HIP backward of a later round is to be held
to these functions the way the forward is held to vz_oracle.py."""
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import vz_oracle as O

PROJECTOR_PREFIX = "model.mm_projector."


def projector_keys(sd: Dict[str, torch.Tensor]):
    """the trainable set of Stage 1, in state-dict order"""
    return [k for k in sd if k.startswith(PROJECTOR_PREFIX)]


def stage1_loss(cfg, sd, input_ids, attention_mask, labels, images, P: O.Prec = O.FP32) -> torch.Tensor:
    """mean token cross-entropy of `forward(input_ids, attention_mask, labels=labels, images=images)`.
    hf:loss/loss_utils.py ForCausalLMLoss: logits -> fp32, shift (logits[:, :-1] vs labels[:, 1:]), ignore_index -100, mean."""
    def encode(imgs, text_emb):
        with torch.no_grad():                                   # ref vision_encoder.py:80 @torch.no_grad()
            feats = O.clip_tower(cfg, sd, imgs, P)
        return O.qformer(cfg, sd, feats, text_emb, P)

    _, pos, mask, _, emb, lab = O.prepare_inputs_labels_for_multimodal(cfg, sd, input_ids, None, attention_mask, None, labels, images,
                                                                        P=P, encode_fn=encode)
    logits, _ = O.llm_forward(cfg, sd, emb, attention_mask=mask, position_ids=pos, P=P)
    V = logits.shape[-1]
    return F.cross_entropy(logits[:, :-1].float().reshape(-1, V), lab[:, 1:].reshape(-1), ignore_index=O.IGNORE_INDEX)


def stage1_grads(cfg, sd, input_ids, attention_mask, labels, images, P: O.Prec = O.FP32) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """(loss, {projector parameter name: dLoss/dparam}) in fp32; every other weight is a constant, as in Stage 1.
    With P = BF16 the forward rounds where the HIP path stores bf16 and - because autograd's backward of a dtype cast casts the
    gradient too - so does the backward: the bf16 band of the gradients."""
    train = dict(sd)
    leaves = {}
    for k in projector_keys(sd):
        leaves[k] = sd[k].detach().clone().requires_grad_(True)
        train[k] = leaves[k]
    with torch.enable_grad():
        loss = stage1_loss(cfg, train, input_ids, attention_mask, labels, images, P)
        loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items()}


def lr_at(step: int, total_steps: int, base_lr: float = 2e-3, warmup_ratio: float = 0.03) -> float:
    """HF `get_cosine_schedule_with_warmup` as the Trainer builds it: linear warm-up over ceil(ratio * total) steps, then half a cosine
    to zero.  `step` counts optimiser steps already taken (0 for the first update)."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, t: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0):
    """one decoupled-weight-decay Adam update (torch.optim.AdamW, no amsgrad), t = 1 for the first step; returns (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v
