#!/usr/bin/env python
"""Pin the CPU oracle (oracle/vz_oracle.py) against the reference itself and emit golden vectors.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where /root/reference exists: it imports
the reference's Python (`vis_zephyr.model.language_model.vis_zephyr.VisZephyrForCausalLM`, which
pulls HF CLIP/Mistral) with the hash-generated weights of vision-zephyr_amd/vz_hip/synth.py loaded
into it, runs every stage of the hot path in fp32 on the CPU, checks the oracle stage by stage
(<= 1e-5 relative to the stage's max magnitude), and writes small fixtures (inputs are regenerated
from seeds; outputs are subsampled slices + sums) to tests/golden/*.npz.  The reference's source
never travels: the GPU box only sees the .npz data and this script's text.

    python oracle/pin_against_reference.py [--out tests/golden] [--llm-layers 2]
"""
import argparse
import json
import os
import sys
import tempfile
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
_OURS = os.path.join(REPO, "vision-zephyr_amd")
sys.path.append(_OURS)                                    # `vz_hip` -> ours (synth only)
sys.path.append(REPO)

import numpy as np
import torch

from vz_hip import synth                                   # noqa: E402
from oracle import vz_oracle as O                          # noqa: E402

# `vis_zephyr` must resolve to the REFERENCE: it has no top-level __init__.py (a namespace package), and the drop-in mirror
# under vision-zephyr_amd/ is a regular package that would win regardless of path order - take our directory off the path
sys.path.remove(_OURS)
sys.path.insert(0, REF)

PINPOINTS = "[[336, 672], [672, 336], [336, 1008], [1008, 336]]"


def build_reference(cfg: synth.ArchConfig, tmp: str):
    from transformers import CLIPVisionConfig, CLIPVisionModel, CLIPImageProcessor
    from vis_zephyr.model.language_model.vis_zephyr import VisZephyrConfig, VisZephyrForCausalLM
    vdir = os.path.join(tmp, "clip")
    vcfg = CLIPVisionConfig(hidden_size=cfg.clip_hidden, intermediate_size=cfg.clip_inter,
                            num_hidden_layers=cfg.clip_layers, num_attention_heads=cfg.clip_heads,
                            image_size=cfg.clip_image, patch_size=cfg.clip_patch, projection_dim=768,
                            hidden_act="quick_gelu", layer_norm_eps=cfg.clip_eps)
    torch.manual_seed(0)
    CLIPVisionModel(vcfg).save_pretrained(vdir)
    CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, resample=3,
                       image_mean=[0.48145466, 0.4578275, 0.40821073],
                       image_std=[0.26862954, 0.26130258, 0.27577711], do_convert_rgb=True).save_pretrained(vdir)
    mcfg = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.n_layers,
                           num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
                           vocab_size=cfg.vocab, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                           sliding_window=cfg.sliding_window, max_position_embeddings=32768,
                           mm_vision_tower=vdir, mm_patch_merge_type="flat", image_aspect_ratio="anyres",
                           mm_grid_pinpoints=PINPOINTS, mm_hidden_size=5120, mm_vision_select_layer="-2,-5,-8,-11,6",
                           mm_vision_select_feature="patch", pad_token_id=2, bos_token_id=1, eos_token_id=2)
    model = VisZephyrForCausalLM(mcfg).eval()
    model.get_vision_tower().load_model()
    model.get_vision_tower().eval()
    return model


def load_synth(model, cfg, seed):
    """Copy hash-generated weights into the reference parameter by parameter; return the oracle's
    weight dict as views of the very same storage."""
    sd = model.state_dict()
    names = set()
    with torch.no_grad():
        out = {}
        for name, t in synth.iter_state_dict(cfg, seed):
            # canonical names follow the reference's pinned transformers 4.52.4 (`...vision_tower.vision_model.*`);
            # the 5.x installed here drops the inner `vision_model.` level
            rname = name if name in sd else name.replace("vision_tower.vision_model.", "vision_tower.")
            assert rname in sd, f"reference has no parameter {name}"
            assert tuple(sd[rname].shape) == tuple(t.shape), (name, sd[rname].shape, t.shape)
            sd[rname].copy_(t)
            names.add(rname)
            out[name] = sd[rname]
    missing = [k for k in sd if k not in names and "position_ids" not in k and "inv_freq" not in k]
    assert not missing, f"synthetic spec misses reference parameters: {missing[:8]}"
    return out


def rel_err(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def sub(t, n=4096):
    """deterministic subsample of a tensor: every k-th flat element."""
    f = t.reshape(-1)
    k = max(1, f.numel() // n)
    return f[::k][:n].double().numpy().astype(np.float32), k


def _round_matrices_in_place(sd):
    """what a bf16 checkpoint holds: every tensor the oracle's Prec.w touches (matrices, embeddings, class token, learned
    queries) rounded to bf16; biases and norm scales stay fp32.  `sd` are views of the reference's own parameters, so the
    reference becomes the W16 model too."""
    n = 0
    for k, t in sd.items():
        if t.ndim >= 2 or k.endswith("class_embedding"):
            t.copy_(t.to(torch.bfloat16).to(torch.float32))
            n += 1
    return n


def _top2(logits):
    v, i = logits.float().topk(2, dim=-1)
    return i.numpy().astype(np.int64), v.numpy().astype(np.float32)


def _case_e(model, cfg, sd, fx, report, check, tag, P, tiles, ids, n_new, t0):
    """BASELINE configs[1]: prefill at S=512 + n_new greedy tokens through the reference's own generate; the oracle is checked
    under teacher forcing with the reference's ids (one forward over prompt + generated ids = the decode steps' logits)."""
    r = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles], None)
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles], P=P)
    check(f"{tag}.splice.embeds", m[4], r[4])
    S = r[4].shape[1]
    g = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, use_cache=True,
                       eos_token_id=None, pad_token_id=2, return_dict_in_generate=True, output_logits=True)
    gen = g.sequences
    assert gen.shape == (1, n_new), gen.shape
    step_ref = torch.stack([x[0].float() for x in g.logits], 0)                      # [n_new, V]
    print(f"[pin] {tag} reference generate done {time.time() - t0:.0f}s ids[:8]={gen[0, :8].tolist()}", flush=True)
    full = torch.cat([m[4], O.embed_tokens(sd, gen[0, :-1], P).unsqueeze(0)], 1)
    _, _, hfin = O.llm_forward(cfg, sd, full, P=P, last_only=True, return_hidden=True)
    step = hfin[0, S - 1:] @ P.w(sd["lm_head.weight"]).t()
    check(f"{tag}.step_logits", step, step_ref, tol=5e-5)
    fx[f"{tag}.generate.ids"] = gen.numpy().astype(np.int64)
    fx[f"{tag}.step_logits.s64"] = step_ref[:, ::64].numpy().astype(np.float32)
    fx[f"{tag}.logits.last"] = step_ref[0].numpy().astype(np.float32)
    fx[f"{tag}.step_top2.ids"], fx[f"{tag}.step_top2.vals"] = _top2(step_ref)
    return m[4], gen, step_ref


def configs2_case(model, cfg, sd, args, t0):
    """bench.py's request (5 tiles seed 1, 1889 ids seed 2 with the image sentinel at 5), W16 reference: last-row logits of the
    prefill, 16 greedy ids with their step logits (subsampled) and top-2 gaps.  The oracle is checked on the prefill's last row
    (its full S=2048 forward) - the per-stage checks at this depth are the --deep run's."""
    n_new = 16
    _round_matrices_in_place(sd)
    tiles = synth.synth_tiles(5, seed=1).to(torch.bfloat16).float()
    ids = synth.synth_ids(1889, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    g = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=n_new, use_cache=True, eos_token_id=None,
                       pad_token_id=2, return_dict_in_generate=True, output_logits=True)
    step_ref = torch.stack([x[0].float() for x in g.logits], 0)
    print(f"[pin] configs[2] reference generate done {time.time() - t0:.0f}s ids={g.sequences[0].tolist()}", flush=True)
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles])
    assert tuple(m[4].shape) == (1, 2048, cfg.hidden)
    lo, _ = O.llm_forward(cfg, sd, m[4], last_only=True)
    e = rel_err(lo[0, -1], step_ref[0])
    print(f"[pin] F16.logits.last rel_err {e:.3e}", flush=True)
    assert e <= 5e-5
    fx = {"F16.generate.ids": g.sequences.numpy().astype(np.int64), "F16.logits.last": step_ref[0].numpy().astype(np.float32),
          "F16.step_logits.s64": step_ref[:, ::64].numpy().astype(np.float32)}
    fx["F16.step_top2.ids"], fx["F16.step_top2.vals"] = _top2(step_ref)
    np.savez_compressed(os.path.join(args.out, f"pin_l{args.llm_layers}_c2.npz"), **fx)
    with open(os.path.join(args.out, f"pin_l{args.llm_layers}_c2.json"), "w") as f:
        json.dump(dict(llm_layers=args.llm_layers, oracle_vs_reference_last_row=e, torch=torch.__version__,
                       transformers=__import__("transformers").__version__,
                       note="BASELINE configs[2] through the reference with bf16-rounded matrices and tiles (W16), fp32 arithmetic, CPU"), f, indent=1)
    print(f"[pin] OK - configs[2] fixtures written ({time.time() - t0:.0f}s)")


def deep_cases(model, cfg, sd, fx, report, check, t0):
    n_new = 128
    tiles_a = synth.synth_tiles(3, seed=1)
    ids_a = synth.synth_ids(32, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    tiles_e = synth.synth_tiles(1, seed=11)
    ids_e = synth.synth_ids(481, cfg.vocab, image_pos=5, seed=12).unsqueeze(0)
    # ---------------- case E, fp32 ----------------
    emb_e, gen_e, step_e = _case_e(model, cfg, sd, fx, report, check, "E", O.FP32, tiles_e, ids_e, n_new, t0)
    # the BF16-policy oracle (activation rounding at the HIP path's store points) on the same teacher-forced ids: the band
    m16 = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids_e, None, None, None, None, [tiles_e], P=O.BF16)
    full = torch.cat([m16[4], O.embed_tokens(sd, gen_e[0, :-1], O.BF16).unsqueeze(0)], 1)
    _, _, hfin = O.llm_forward(cfg, sd, full, P=O.BF16, last_only=True, return_hidden=True)
    step16 = hfin[0, emb_e.shape[1] - 1:] @ O.BF16.w(sd["lm_head.weight"]).t()
    fx["E.bf16_oracle.step_logits.s64"] = step16[:, ::64].numpy().astype(np.float32)
    fx["E.bf16_oracle.argmax"] = step16.argmax(-1).numpy().astype(np.int64)
    print(f"[pin] E bf16-policy oracle vs fp32 reference: rel-L2 "
          f"{float((step16 - step_e).norm() / step_e.norm()):.3e} ({time.time() - t0:.0f}s)", flush=True)
    # ---------------- W16: the reference itself on bf16-rounded matrices and tiles ----------------
    lo32_a = model(input_ids=ids_a, images=[tiles_a]).logits
    n = _round_matrices_in_place(sd)
    print(f"[pin] rounded {n} tensors of the reference to bf16 in place ({time.time() - t0:.0f}s)", flush=True)
    ta = tiles_a.to(torch.bfloat16).float()
    te = tiles_e.to(torch.bfloat16).float()
    lo_ref = model(input_ids=ids_a, images=[ta]).logits
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids_a, None, None, None, None, [ta])   # weights are rounded already
    lo, _ = O.llm_forward(cfg, sd, m[4])
    check("A16.logits", lo, lo_ref, tol=5e-5)
    fx["A16.logits.last"] = lo_ref[0, -1].numpy().astype(np.float32)
    report["A.weight_rounding_rel_l2"] = float((lo_ref - lo32_a).norm() / lo32_a.norm())
    print(f"[pin] A: weight rounding alone moves the logits by rel-L2 {report['A.weight_rounding_rel_l2']:.3e}", flush=True)
    gen_a = model.generate(input_ids=ids_a, images=[ta], do_sample=False, max_new_tokens=6, use_cache=True,
                           eos_token_id=None, pad_token_id=2)
    fx["A16.generate.ids"] = gen_a.numpy().astype(np.int64)
    tower = model.get_vision_tower()
    hs_ref = tower.vision_tower(ta, output_hidden_states=True)["hidden_states"]
    hs = O.clip_hidden_states(cfg, sd, ta)
    check("A16.clip.hs24", hs[24], hs_ref[24])
    fused = tower(ta)
    check("A16.fused", O.fusion(cfg, hs), fused)
    text_ids = ids_a[0][ids_a[0] != O.IMAGE_TOKEN_INDEX]
    te_a = model.get_model().embed_tokens(text_ids).unsqueeze(0).expand(3, -1, -1)
    check("A16.encode_images", O.qformer(cfg, sd, O.fusion(cfg, hs), te_a), model.encode_images(ta, te_a))
    _, gen16, step16_ref = _case_e(model, cfg, sd, fx, report, check, "E16", O.FP32, te, ids_e, n_new, t0)
    # W16 logits on the fp32 run's ids (teacher forcing with E.generate.ids) so FP32 and W16 are comparable step by step
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids_e, None, None, None, None, [te])
    full = torch.cat([m[4], O.embed_tokens(sd, gen_e[0, :-1], O.FP32).unsqueeze(0)], 1)
    _, _, hfin = O.llm_forward(cfg, sd, full, last_only=True, return_hidden=True)
    stepw = hfin[0, m[4].shape[1] - 1:] @ sd["lm_head.weight"].t()
    fx["E.w16_on_fp32_ids.step_logits.s64"] = stepw[:, ::64].numpy().astype(np.float32)
    report["E.weight_rounding_rel_l2"] = float((stepw - step_e).norm() / step_e.norm())
    report["E.first_id_divergence_fp32_vs_w16"] = int((gen16[0] != gen_e[0]).nonzero()[0]) if bool((gen16 != gen_e).any()) else -1
    print(f"[pin] E: weight rounding alone moves the step logits by rel-L2 {report['E.weight_rounding_rel_l2']:.3e}; "
          f"fp32 and W16 greedy ids first differ at step {report['E.first_id_divergence_fp32_vs_w16']}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--llm-layers", type=int, default=2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--deep", action="store_true",
                    help="also pin case E (BASELINE configs[1]: 1 tile + 481 ids -> S=512, 128 greedy tokens) and re-run cases "
                         "A and E through the reference with its matrices rounded to bf16 in place (the W16 model)")
    ap.add_argument("--configs2", action="store_true",
                    help="ONLY BASELINE configs[2] (5 tiles + 1889 ids -> S=2048, 16 greedy tokens) through the W16 reference "
                         "(matrices rounded to bf16 in place) -> pin_l<k>_c2.npz: what bench.py checks its own request against")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_grad_enabled(False)
    cfg = synth.ArchConfig(n_layers=args.llm_layers)
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        model = build_reference(cfg, tmp)
    print(f"[pin] reference built in {time.time() - t0:.0f}s", flush=True)
    sd = load_synth(model, cfg, args.seed)
    print(f"[pin] weights loaded {time.time() - t0:.0f}s", flush=True)
    report = {}
    fx = {}
    if args.configs2:
        return configs2_case(model, cfg, sd, args, t0)

    def check(name, mine, ref, tol=1e-5):
        e = rel_err(mine, ref)
        report[name] = e
        print(f"[pin] {name:34s} rel_err {e:.3e}  shape {tuple(ref.shape)}", flush=True)
        assert e <= tol, f"oracle deviates from the reference at {name}: {e}"
        s, k = sub(ref)
        fx[name + ".sub"] = s
        fx[name + ".stride"] = np.int64(k)
        fx[name + ".sum"] = np.float64(ref.double().sum())
        fx[name + ".abssum"] = np.float64(ref.double().abs().sum())
        fx[name + ".shape"] = np.array(ref.shape, dtype=np.int64)

    # ---------------- case A: C1 shape - 3 tiles, 32 ids, one image sentinel at index 5 ----------------
    tiles = synth.synth_tiles(3, seed=1)
    ids = synth.synth_ids(32, cfg.vocab, image_pos=5, seed=2).unsqueeze(0)
    tower = model.get_vision_tower()
    out = tower.vision_tower(tiles, output_hidden_states=True)
    hs_ref = out["hidden_states"]
    hs = O.clip_hidden_states(cfg, sd, tiles)
    assert len(hs_ref) == len(hs) == cfg.clip_layers + 1
    for i in (0, 1, 4, 12, 24):
        check(f"A.clip.hs{i}", hs[i], hs_ref[i])
    fused_ref = tower(tiles)
    fused = O.fusion(cfg, hs)
    check("A.fused", fused, fused_ref)
    text_ids = ids[0][ids[0] != O.IMAGE_TOKEN_INDEX]
    te_ref = model.get_model().embed_tokens(text_ids).unsqueeze(0).expand(3, -1, -1)
    te = O.embed_tokens(sd, text_ids, O.FP32).unsqueeze(0).expand(3, -1, -1)
    qf = model.get_model().mm_projector
    # per-block outputs of the reference Q-Former
    f_n = qf.pre_norm(fused_ref)
    x = torch.cat([qf.learned_queries.unsqueeze(0).expand(3, -1, -1), te_ref], 1)
    x = qf.blocks[0](x, f_n)[:, :32]
    ref_blocks = [x]
    for blk in qf.blocks[1:]:
        x = blk(x, f_n)
        ref_blocks.append(x)
    proj_ref = qf(fused_ref, text_embeddings=te_ref)
    proj, blocks = O.qformer(cfg, sd, fused, te, return_blocks=True)
    for i in (0, 1, 7):
        check(f"A.qformer.block{i}", blocks[i], ref_blocks[i])
    check("A.qformer.out", proj, proj_ref)
    check("A.encode_images", O.encode_images(cfg, sd, tiles, te), model.encode_images(tiles, te_ref))
    r = model.prepare_inputs_labels_for_multimodal(ids, None, None, None, None, [tiles], None)
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids, None, None, None, None, [tiles])
    assert r[0] is None and m[0] is None and r[1] is None and m[1] is None and r[2] is None and m[2] is None
    assert r[5] is None and m[5] is None
    check("A.splice.embeds", m[4], r[4])
    assert tuple(r[4].shape) == (1, 31 + 32 * 3, cfg.hidden)
    lo_ref = model(input_ids=ids, images=[tiles]).logits
    lo, _ = O.llm_forward(cfg, sd, m[4])
    check("A.logits", lo, lo_ref, tol=2e-5)
    fx["A.logits.last"] = lo_ref[0, -1].numpy().astype(np.float32)
    gen_ref = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=6, use_cache=True,
                             eos_token_id=None, pad_token_id=2)
    gen, glog = O.greedy_generate(cfg, sd, m[4], 6, return_logits=True)
    print("[pin] A.generate ref", gen_ref.tolist(), "oracle", gen.tolist())
    assert gen_ref.shape == (1, 6) and torch.equal(gen_ref, gen)
    fx["A.generate.ids"] = gen_ref.numpy().astype(np.int64)
    fx["A.generate.step_logits.sub"] = glog[0, :, ::37].numpy().astype(np.float32)

    # ---------------- case B: batch of 2, unequal lengths + padding mask, labels, 5-D-equivalent list ----------------
    tb0 = synth.synth_tiles(2, seed=3)
    tb1 = synth.synth_tiles(1, seed=4)
    ids_b = torch.full((2, 20), 2, dtype=torch.long)
    a = synth.synth_ids(20, cfg.vocab, image_pos=1, seed=5)
    b = synth.synth_ids(13, cfg.vocab, image_pos=7, seed=6)
    ids_b[0] = a
    ids_b[1, :13] = b
    mask_b = torch.zeros(2, 20, dtype=torch.long)
    mask_b[0] = 1
    mask_b[1, :13] = 1
    pos_b = torch.arange(20).unsqueeze(0).expand(2, -1).contiguous()
    lab_b = ids_b.clone()
    lab_b[ids_b == O.IMAGE_TOKEN_INDEX] = O.IGNORE_INDEX
    r = model.prepare_inputs_labels_for_multimodal(ids_b, pos_b, mask_b, None, lab_b, [tb0, tb1], None)
    m = O.prepare_inputs_labels_for_multimodal(cfg, sd, ids_b, pos_b, mask_b, None, lab_b, [tb0, tb1])
    check("B.splice.embeds", m[4], r[4])
    assert torch.equal(r[1], m[1]) and torch.equal(r[2], m[2]) and torch.equal(r[5], m[5])
    assert r[2].dtype == m[2].dtype
    fx["B.position_ids"] = r[1].numpy()
    fx["B.attention_mask"] = r[2].numpy()
    fx["B.labels"] = r[5].numpy()
    ob = model(input_ids=ids_b, attention_mask=mask_b, position_ids=pos_b, labels=lab_b, images=[tb0, tb1])
    lo_b, _ = O.llm_forward(cfg, sd, m[4], attention_mask=m[2], position_ids=m[1])
    valid = m[2].bool()
    check("B.logits.valid", lo_b[valid], ob.logits[valid], tol=2e-5)
    fx["B.loss"] = np.float64(ob.loss)

    # ---------------- case C: text-only generate ----------------
    ids_c = synth.synth_ids(9, cfg.vocab, image_pos=-1, seed=7).unsqueeze(0)
    gen_ref = model.generate(input_ids=ids_c, images=None, do_sample=False, max_new_tokens=4, use_cache=True,
                             eos_token_id=None, pad_token_id=2)
    gen = O.generate(cfg, sd, ids_c, None, 4)
    print("[pin] C.generate ref", gen_ref.tolist(), "oracle", gen.tolist())
    assert torch.equal(gen_ref, gen)
    fx["C.generate.ids"] = gen_ref.numpy().astype(np.int64)

    # ---------------- case D: mm_vision_select_feature = 'cls_patch' (ref vision_encoder.py:66-73): 577 tokens per tile ----------------
    tower.select_feature = "cls_patch"
    try:
        tiles_d = synth.synth_tiles(2, seed=8)
        fused_ref_d = tower(tiles_d)
        assert tuple(fused_ref_d.shape) == (2, cfg.clip_tokens, 5 * cfg.clip_hidden)
        fused_d = O.clip_tower(cfg, sd, tiles_d, select_feature="cls_patch")
        check("D.fused.cls_patch", fused_d, fused_ref_d)
        te_d = te[:2]
        check("D.encode_images.cls_patch", O.qformer(cfg, sd, fused_d, te_d), model.encode_images(tiles_d, te_ref[:2]))
    finally:
        tower.select_feature = "patch"

    if args.deep:
        deep_cases(model, cfg, sd, fx, report, check, t0)

    meta = dict(llm_layers=args.llm_layers, seed=args.seed, report=report,
                torch=torch.__version__, transformers=__import__("transformers").__version__,
                note="outputs of the reference (fp32, CPU) on hash-generated weights; inputs regenerate from seeds "
                     "(A: tiles seed 1 n=3, ids seed 2 n=32 image_pos 5; B: tiles seeds 3 (n=2) / 4 (n=1), ids seeds 5/6; "
                     "C: ids seed 7 n=9; D: tiles seed 8 n=2 with the text of case A, select_feature cls_patch)")
    np.savez_compressed(os.path.join(args.out, f"pin_l{args.llm_layers}.npz"), **fx)
    with open(os.path.join(args.out, f"pin_l{args.llm_layers}.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"[pin] OK - all stages within tolerance; fixtures written ({time.time() - t0:.0f}s)")


if __name__ == "__main__":
    main()
