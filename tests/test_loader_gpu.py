"""`load_pretrained_model` end to end on the GPU (ref:vis_zephyr/model/builder.py:16-161): a checkpoint laid out
the way the reference's loader expects it - Zephyr backbone shards in `model_base`, `config.json` + `mm_projector.bin` in
`model_path`, an HF CLIP directory named by `mm_vision_tower` - holding the hash-generated weights, so the loaded model can be
compared bit for bit with `from_synthetic`."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Tok:            # stands in for the sentencepiece tokenizer (no tokenizer files offline): what the builder uses of it
    def __init__(self, n):
        self.n = n
        self.added = []

    def add_tokens(self, toks, special_tokens=False):
        self.added += list(toks)
        self.n += len(toks)
        return len(toks)

    def __len__(self):
        return self.n


@pytest.mark.parametrize("mode", ["bf16", "8bit", "4bit"])
def test_load_pretrained_model_matches_from_synthetic(tmp_path, monkeypatch, mode):
    load_8bit, load_4bit = mode == "8bit", mode == "4bit"
    from safetensors.torch import save_file
    import transformers
    from vz_hip import synth
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    from vis_zephyr.model.builder import load_pretrained_model
    cfg = synth.ArchConfig(n_layers=1, vocab=300)
    base, ckpt, clip = tmp_path / "zephyr-7b-beta", tmp_path / "vis-zephyr-7b-v1-pretrain", tmp_path / "clip-vit-large-patch14-336"
    for d in (base, ckpt, clip):
        d.mkdir()
    llm, vit, proj = {}, {}, {}
    for k, v in synth.iter_state_dict(cfg, 0, device="cuda"):
        # matrices as a bf16 checkpoint stores them; vectors (biases, norm scales) in fp32, which is how the engine keeps them - so the
        # loaded engine holds exactly the bytes from_synthetic produces
        v = (v.to(torch.bfloat16) if v.dim() >= 2 else v.float()).cpu().contiguous()
        if k.startswith("model.vision_tower.vision_tower."):
            vit[k[len("model.vision_tower.vision_tower."):]] = v
        elif k.startswith("model.mm_projector."):
            proj[k] = v
        else:
            llm[k] = v
    keys = sorted(llm)
    save_file({k: llm[k] for k in keys[: len(keys) // 2]}, str(base / "model-00001-of-00002.safetensors"))
    save_file({k: llm[k] for k in keys[len(keys) // 2:]}, str(base / "model-00002-of-00002.safetensors"))
    save_file(vit, str(clip / "model.safetensors"))
    json.dump({"crop_size": {"height": 336, "width": 336}, "size": {"shortest_edge": 336}, "image_mean": [0.48145466, 0.4578275, 0.40821073],
               "image_std": [0.26862954, 0.26130258, 0.27577711], "image_processor_type": "CLIPImageProcessor", "resample": 3,
               "do_resize": True, "do_center_crop": True, "do_normalize": True, "do_rescale": True, "do_convert_rgb": True},
              open(clip / "preprocessor_config.json", "w"))
    torch.save(proj, str(ckpt / "mm_projector.bin"))
    json.dump({"model_type": "vis_zephyr", "architectures": ["VisZephyrForCausalLM"], "hidden_size": cfg.hidden, "intermediate_size": cfg.inter,
               "num_hidden_layers": 1, "num_attention_heads": cfg.n_heads, "num_key_value_heads": cfg.n_kv_heads, "vocab_size": 300,
               "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "sliding_window": 4096, "mm_vision_tower": str(clip),
               "mm_hidden_size": 5120, "mm_patch_merge_type": "flat", "image_aspect_ratio": "anyres", "mm_vision_select_feature": "patch",
               "mm_vision_select_layer": "-2,-5,-8,-11,6", "mm_projector_type": "mlp2x_gelu", "eos_token_id": 2, "pad_token_id": 2,
               "bos_token_id": 1}, open(ckpt / "config.json", "w"))
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: _Tok(300)))
    tok, model, proc, ctx = load_pretrained_model(str(ckpt), str(base), "vis-zephyr-7b-v1-pretrain", load_8bit=load_8bit, load_4bit=load_4bit, max_ctx=256)
    assert ctx == 2048 and len(tok) == 301 and tok.added == ["<im_patch>"]        # ref builder.py:141-153: vocabulary grows by one
    assert model.config.vocab_size == 301 and model.engine.cfg.vocab == 301 and model.engine.weight_fp8 == load_8bit and model.engine.weight_nf4 == load_4bit
    assert proc is not None and proc.crop_size["height"] == 336
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=1, num_attention_heads=cfg.n_heads,
                         num_key_value_heads=cfg.n_kv_heads, vocab_size=300, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                         sliding_window=4096, eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = str(clip)
    hf.mm_patch_merge_type = "flat"
    hf.mm_hidden_size = 5120
    ref = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=2, max_text=64, weight_fp8=load_8bit, weight_nf4=load_4bit)
    tiles = synth.synth_tiles(2, seed=1).to(model.device).bfloat16()
    ids = synth.synth_ids(20, 300, image_pos=3, seed=2).unsqueeze(0).to(model.device)
    diff = [n for n, t in ref.engine.w.items() if not n.startswith(("llm.embed", "llm.lm_head")) and not torch.equal(t, model.engine.w[n])]
    assert not diff, f"engine tensors differ after loading: {diff[:8]} ({len(diff)} of {len(ref.engine.w)})"
    a = model(input_ids=ids, images=[tiles]).logits
    b = ref(input_ids=ids, images=[tiles]).logits
    # same weights and kernels up to the lm_head GEMM, whose split-K form needs N % 4 == 0 (300 rows: yes, 301: no): the fp32
    # logits agree to re-association of the K sum
    assert a.shape[-1] == 301 and float((a[..., :300] - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # the added row is the mean of the old lm_head rows (what resize_token_embeddings gives; vz_hip/weights.py::resize_vocab)
    if not load_8bit:
        lm = ref.engine.w["llm.lm_head"].float()
        assert torch.equal(model.engine.w["llm.lm_head"][300].float(), lm[:300].mean(0).bfloat16().float())
    out = model.generate(input_ids=ids, images=[tiles], do_sample=False, max_new_tokens=4, eos_token_id=None)
    assert out.shape == (1, 4)


def _write_checkpoint(tmp_path, cfg, synth, vocab):
    """the same three-directory layout as above, returned as paths; the CLIP directory also sits in an HF-cache layout
    (models--openai--clip-vit-large-patch14-336/snapshots/<rev>) so hub ids can be resolved offline."""
    from safetensors.torch import save_file
    base, ckpt = tmp_path / "zephyr-7b-beta", tmp_path / "vis-zephyr-7b-v1-pretrain"
    cache = tmp_path / "hfcache"
    clip = cache / "models--openai--clip-vit-large-patch14-336" / "snapshots" / "0000000000000000000000000000000000000000"
    zeph = cache / "models--HuggingFaceH4--zephyr-7b-beta" / "snapshots" / "1111111111111111111111111111111111111111"
    for d in (base, ckpt, clip, zeph):
        d.mkdir(parents=True)
    llm, vit, proj = {}, {}, {}
    for k, v in synth.iter_state_dict(cfg, 0, device="cuda"):
        v = (v.to(torch.bfloat16) if v.dim() >= 2 else v.float()).cpu().contiguous()
        if k.startswith("model.vision_tower.vision_tower."):
            vit[k[len("model.vision_tower.vision_tower."):]] = v
        elif k.startswith("model.mm_projector."):
            proj[k] = v
        else:
            llm[k] = v
    save_file(llm, str(base / "model.safetensors"))
    save_file(llm, str(zeph / "model.safetensors"))
    save_file(vit, str(clip / "model.safetensors"))
    torch.save(proj, str(ckpt / "mm_projector.bin"))
    conf = {"model_type": "vis_zephyr", "architectures": ["VisZephyrForCausalLM"], "hidden_size": cfg.hidden, "intermediate_size": cfg.inter,
            "num_hidden_layers": cfg.n_layers, "num_attention_heads": cfg.n_heads, "num_key_value_heads": cfg.n_kv_heads, "vocab_size": vocab,
            "rms_norm_eps": cfg.rms_eps, "rope_theta": cfg.rope_theta, "sliding_window": 4096,
            "mm_vision_tower": "openai/clip-vit-large-patch14-336",                   # a hub id, as the shipped config.json:23 has it
            "mm_hidden_size": 5120, "mm_patch_merge_type": "flat", "image_aspect_ratio": "anyres", "mm_vision_select_feature": "patch",
            "mm_vision_select_layer": "-2,-5,-8,-11,6", "mm_projector_type": "mlp2x_gelu", "eos_token_id": 2, "pad_token_id": 2, "bos_token_id": 1}
    for d in (ckpt, base, zeph):
        json.dump(conf, open(d / "config.json", "w"))
    return base, ckpt, cache


def test_from_pretrained_auto_classes_and_hub_cache(tmp_path, monkeypatch):
    """a1: `AutoConfig` / `AutoModelForCausalLM` registration (ref:vis_zephyr/model/language_model/vis_zephyr.py:173-174),
    `VisZephyrForCausalLM.from_pretrained(model_base, config=cfg_pretrained)` + `load_state_dict(mm_projector, strict=False)`
    + `get_vision_tower().load_model()` exactly as ref:vis_zephyr/model/builder.py:102-138 strings them together, with the
    hub ids of script/run_cli.sh / config.json resolved through a LOCAL HF cache (`local_files_only`)."""
    import transformers
    from transformers import AutoConfig, AutoModelForCausalLM
    from vz_hip import synth
    from vz_hip import weights as W
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    from vis_zephyr.model.builder import load_pretrained_model
    cfg = synth.ArchConfig(n_layers=1, vocab=300)
    base, ckpt, cache = _write_checkpoint(tmp_path, cfg, synth, 300)
    monkeypatch.setenv("HF_HUB_CACHE", str(cache))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    import huggingface_hub.constants as hc
    monkeypatch.setattr(hc, "HF_HUB_CACHE", str(cache), raising=False)
    assert W.resolve_hub_path("openai/clip-vit-large-patch14-336").startswith(str(cache))
    with pytest.raises(FileNotFoundError):
        W.resolve_hub_path("nobody/not-in-the-cache")
    cfg_pretrained = AutoConfig.from_pretrained(str(ckpt))
    assert isinstance(cfg_pretrained, VisZephyrConfig) and cfg_pretrained.mm_vision_tower == "openai/clip-vit-large-patch14-336"
    # the reference's base + projector sequence, the base named by its hub id
    model = AutoModelForCausalLM.from_pretrained("HuggingFaceH4/zephyr-7b-beta", config=cfg_pretrained, low_cpu_mem_usage=True,
                                                 max_ctx=256, max_tiles=2, max_text=64)
    assert isinstance(model, VisZephyrForCausalLM) and not model.engine.ready
    res = model.load_state_dict(torch.load(str(ckpt / "mm_projector.bin"), map_location="cpu"), strict=False)
    assert res.unexpected_keys == []
    tower = model.get_vision_tower()
    assert not tower.is_loaded
    tower.load_model()                                      # streams the CLIP weights out of the cached snapshot
    assert tower.is_loaded and tower.image_processor is not None
    hf = VisZephyrConfig(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=1, num_attention_heads=cfg.n_heads,
                         num_key_value_heads=cfg.n_kv_heads, vocab_size=300, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                         sliding_window=4096, eos_token_id=2, pad_token_id=2, bos_token_id=1)
    hf.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    hf.mm_patch_merge_type = "flat"
    ref = VisZephyrForCausalLM.from_synthetic(hf, seed=0, max_batch=1, max_ctx=256, max_tiles=2, max_text=64)
    tiles = synth.synth_tiles(2, seed=1).to(model.device).bfloat16()
    ids = synth.synth_ids(20, 300, image_pos=3, seed=2).unsqueeze(0).to(model.device)
    a = model(input_ids=ids, images=[tiles]).logits             # first use finalizes the engine
    assert model.engine.ready and torch.equal(a, ref(input_ids=ids, images=[tiles]).logits)
    # a model whose projector never arrives fails loudly, naming the tensor
    bare = VisZephyrForCausalLM.from_pretrained(str(base), config=cfg_pretrained, max_ctx=128)
    with pytest.raises(RuntimeError, match="never registered"):
        bare(input_ids=ids)
    # load_pretrained_model with every path given as a hub id / cached snapshot (script/run_cli.sh's form)
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(lambda *a, **k: _Tok(300)))
    tok, m2, proc, ctx = load_pretrained_model(str(ckpt), "HuggingFaceH4/zephyr-7b-beta", "vis-zephyr-7b-v1-pretrain", max_ctx=128)
    assert m2.engine.cfg.vocab == 301 and ctx == 2048
