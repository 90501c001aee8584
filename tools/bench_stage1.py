#!/usr/bin/env python
"""BASELINE.json configs[3] (Stage-1 pretrain shape, SURVEY section 8(d) "C4") as a forward-only measurement on ONE GPU without the
tensor parallelism: `forward` over 64 plain-caption samples x 5 anyres tiles (320 tiles), 40 ids each -> S = 39 + 160 = 199 rows
per sample, labels in, loss + logits for all positions out.

    python tools/bench_stage1.py [samples] [layers]            forward only
    python tools/bench_stage1.py [samples] [layers] train [mb]  the Stage-1 OPTIMISER STEP: forward with saved
                                                                activations + backward (frozen Zephyr input gradients, all 165
                                                                projector gradients) + AdamW, in micro-batches of mb samples (16)

Prints the two halves (tiles -> visual tokens -> spliced embeddings; Zephyr forward + loss) and the algorithmic TFLOP/s of each
(CLIP 381.9 GFLOP per tile, Q-Former 59.4 x 8 GFLOP per tile for its 32-row blocks + the cross-attention K/V projections,
Zephyr 2 x weights x rows + causal attention + lm_head on every row: SURVEY section 8(d))."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "vision-zephyr_amd"))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from vz_hip import synth  # noqa: E402
from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM  # noqa: E402

from vz_hip import binding as _B  # noqa: E402
for kv in filter(None, os.environ.get("VZ_TUNE", "").split(",")):       # experiments: "knob=value,..." for vz_tune_set (32=0: materialising attention backward)
    _k, _v = (int(t) for t in kv.split("="))
    _B.check(_B.lib().vz_tune_set(_k, _v))
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
Bn = nums[0] if nums else 64
layers = nums[1] if len(nums) > 1 else 32
TRAIN = "train" in sys.argv[1:]
MB = nums[2] if len(nums) > 2 else 16
N_TILES, N_IDS = 5, 40
S = N_IDS - 1 + 32 * N_TILES

hf = VisZephyrConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32, num_key_value_heads=8,
                     vocab_size=32000, rms_norm_eps=1e-5, sliding_window=4096, eos_token_id=2, pad_token_id=2, bos_token_id=1)
hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
hf.mm_patch_merge_type = "flat"
hf.image_aspect_ratio = "anyres"
hf.mm_grid_pinpoints = "[[336, 672], [672, 336], [336, 1008], [1008, 336], [672, 672]]"
hf.mm_hidden_size = 5120
MAXB = min(Bn, max(16, MB)) if TRAIN else min(Bn, 16)
model = VisZephyrForCausalLM.from_synthetic(hf, seed=0, device="cuda:0", max_batch=MAXB, max_ctx=S + 8, max_tiles=N_TILES * MAXB,
                                            max_text=N_IDS + 8)
dev = model.device
ids = torch.stack([synth.synth_ids(N_IDS, 32000, image_pos=1, seed=100 + b) for b in range(Bn)]).to(dev)   # plain: image after BOS
labels = ids.clone()
labels[:, :2] = -100
tiles = [synth.synth_tiles(N_TILES, seed=300 + b).to(dev).bfloat16() for b in range(Bn)]
mask = torch.ones_like(ids)


def run():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, pos, m2, _, emb, lab = model.prepare_inputs_labels_for_multimodal(ids, None, mask, None, labels, tiles)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = model(inputs_embeds=emb, attention_mask=m2, position_ids=pos, labels=lab)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert tuple(out.logits.shape) == (Bn, S, 32000) and bool(torch.isfinite(out.loss))
    return t1 - t0, t2 - t1


T = Bn * N_TILES
if TRAIN:
    from vz_hip.train import Stage1Trainer
    tr = Stage1Trainer(model)
    arch = model.arch
    per_layer_w = ((arch.n_heads + 2 * arch.n_kv_heads) * arch.head_dim + arch.hidden + 3 * arch.inter) * arch.hidden
    f_clip, f_qf = 381.9e9 * T, (8 * 59.4e9) * T
    f_lin = Bn * (2 * S * layers * per_layer_w + 2 * S * arch.vocab * arch.hidden)
    f_att = Bn * layers * 4 * S * S * arch.hidden / 2
    # step = forward + Zephyr input gradients (1 x its linears, 2.5 x its attention) + Q-Former backward (2 x: weight + input gradients)
    f_step = f_clip + 3 * f_qf + 2 * f_lin + 3.5 * f_att
    losses, times = [], []
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses.append(tr.step(ids, mask, labels, tiles, lr=2e-3, micro_batch=MB))      # --mm_projector_lr (ref:script/pretrain.sh:16)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    print(f"stage-1 optimiser step: {Bn} samples x {N_TILES} tiles, S={S}, {layers} layers, micro-batches of {MB}, one GPU")
    print(f"  losses over 4 steps on the same batch: {' '.join(f'{v:.4f}' for v in losses)}")
    print(f"  step time: {best * 1e3:8.1f} ms  ({Bn / best:6.1f} samples/s, {f_step / best / 1e12:6.1f} TFLOP/s algorithmic = {f_step / best / 2.5e15 * 100:4.1f} % of 2.5 PF; "
          f"first step incl. allocations {times[0] * 1e3:.0f} ms)")
    free, total = torch.cuda.mem_get_info()
    print(f"  device memory in use after the steps: {(total - free) / 2**30:.1f} GiB of {total / 2**30:.0f}")
    sys.exit(0)
run()
enc, llm = min((run() for _ in range(3)), key=sum)
arch = model.arch
per_layer_w = ((arch.n_heads + 2 * arch.n_kv_heads) * arch.head_dim + arch.hidden + 3 * arch.inter) * arch.hidden
f_clip = 381.9e9 * T
f_qf = (8 * 59.4e9) * T                                             # 32-row blocks incl. the K/V projections of the 576 visual rows
f_llm = Bn * (2 * S * layers * per_layer_w + 2 * S * arch.vocab * arch.hidden + layers * 4 * S * S * arch.hidden / 2)
print(f"stage-1 shape: {Bn} samples x {N_TILES} tiles, S={S}, {layers} layers, one GPU, forward only")
print(f"  tiles -> spliced embeddings: {enc * 1e3:8.1f} ms  ({T / enc:7.1f} tiles/s, {(f_clip + f_qf) / enc / 1e12:6.1f} TFLOP/s algorithmic)")
print(f"  Zephyr forward + loss:       {llm * 1e3:8.1f} ms  ({Bn * S / llm:7.0f} rows/s,  {f_llm / llm / 1e12:6.1f} TFLOP/s algorithmic)")
print(f"  whole forward:               {(enc + llm) * 1e3:8.1f} ms  ({Bn / (enc + llm):6.1f} samples/s, "
      f"{(f_clip + f_qf + f_llm) / (enc + llm) / 1e12:6.1f} TFLOP/s = {(f_clip + f_qf + f_llm) / (enc + llm) / 2.5e15 * 100:4.1f} % of 2.5 PF)")
