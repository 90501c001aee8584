"""Host-side mirror of the reference's feeder modules against the known answers captured from the reference
 and the loader's file handling on small synthetic checkpoints."""
import json
import os

import pytest
import torch
from PIL import Image


def test_select_best_fit_resolution_known_answers():
    from vis_zephyr.model.multi_scale_process import calculate_grid_shape, select_best_fit_resolution
    pins = [[336, 672], [672, 336], [336, 1008], [1008, 336]]
    cases = {(637, 336): (672, 336), (681, 336): (1008, 336), (1920, 804): (1008, 336), (336, 336): (336, 672),
             (640, 480): (672, 336), (480, 640): (336, 672), (300, 1000): (336, 1008)}
    for size, want in cases.items():
        assert tuple(select_best_fit_resolution(size, pins)) == want, size
    pins5 = pins + [[672, 672]]
    assert tuple(select_best_fit_resolution((672, 672), pins5)) == (672, 672)
    assert tuple(select_best_fit_resolution((640, 480), pins5)) == (672, 672)
    assert calculate_grid_shape((1920, 804), "'[[336, 672], [672, 336], [336, 1008], [1008, 336]]'", 336) == (3, 1)
    with pytest.raises(ValueError):
        calculate_grid_shape((10, 10), "7", 336)


class _Proc:
    crop_size = {"height": 336, "width": 336}
    image_mean = [0.48145466, 0.4578275, 0.40821073]

    def preprocess(self, img, return_tensors="pt"):
        assert img.size == (336, 336)
        t = torch.from_numpy(__import__("numpy").asarray(img.convert("RGB"))).permute(2, 0, 1).float() / 255.0
        return {"pixel_values": t.unsqueeze(0)}

    def __call__(self, images, return_tensors="pt"):
        return {"pixel_values": torch.cat([self.preprocess(i.resize((336, 336)))["pixel_values"] for i in images], 0)}


def test_anyres_tile_count_and_order():
    from vis_zephyr.model.multi_scale_process import process_any_resolution_image, resize_pad_image, divide_to_patches
    img = Image.new("RGB", (637, 336), (255, 0, 0))
    pins = "[[336, 672], [672, 336], [336, 1008], [1008, 336]]"
    tiles = process_any_resolution_image(img, _Proc(), pins)
    assert tuple(tiles.shape) == (3, 3, 336, 336)                 # global + 2 crops (Appendix B)
    assert process_any_resolution_image(Image.new("RGB", (681, 336)), _Proc(), pins).shape[0] == 4
    assert process_any_resolution_image(Image.new("RGB", (640, 480)), _Proc(), pins[:-1] + ", [672, 672]]").shape[0] == 5
    padded = resize_pad_image(Image.new("RGB", (100, 50), (0, 255, 0)), (336, 672))
    assert padded.size == (336, 672) and padded.getpixel((0, 0)) == (0, 0, 0) and padded.getpixel((168, 336)) == (0, 255, 0)
    crops = divide_to_patches(Image.new("RGB", (672, 336)), 336)
    assert len(crops) == 2 and crops[0].size == (336, 336)


class _Tok:
    bos_token_id = 1

    def __call__(self, text):
        return type("E", (), {"input_ids": [1] + [ord(c) for c in text]})()


def test_tokenizer_image_token_and_prompt():
    from vis_zephyr.model.mm_utils import KeywordsStoppingCriteria, get_model_name_from_path, tokenizer_image_token
    from vis_zephyr.conversation import SeparatorStyle, templates
    ids = tokenizer_image_token("ab<image>cd<image>ef", _Tok())
    assert ids == [1, 97, 98, -200, 99, 100, -200, 101, 102]
    assert tokenizer_image_token("<image>\nhi", _Tok(), return_tensors="pt").tolist() == [1, -200, 10, 104, 105]
    with pytest.raises(ValueError):
        tokenizer_image_token("x", _Tok(), return_tensors="np")
    assert get_model_name_from_path("/a/b/vis-zephyr/checkpoint-12/") == "vis-zephyr_checkpoint-12"
    conv = templates["zephyr_v1"].copy()
    conv.append_message(conv.roles[0], "<image>\nWhat is this?")
    conv.append_message(conv.roles[1], None)
    p = conv.get_prompt()
    assert p.startswith("<|system|>\nYou are an AI assistant") and p.endswith("</s><|user|>\n<image>\nWhat is this?</s><|assistant|>\n")
    assert templates["plain"].separator_style == SeparatorStyle.PLAIN
    with pytest.raises(ValueError):
        templates["plain"].get_prompt()
    stop = KeywordsStoppingCriteria(["</s>"], _Tok(), torch.zeros(1, 2, dtype=torch.long))
    kw = [ord(c) for c in "</s>"]
    assert not stop(torch.tensor([[5, 6, 7]]), None)
    assert stop(torch.tensor([[5, 6] + kw]), None)
    assert not stop(torch.tensor([kw[-2:] + [9]]), None)


def test_checkpoint_stream_lora_merge_and_vocab_resize(tmp_path):
    from safetensors.torch import save_file
    from vz_hip import weights as W
    base, ckpt, clip = tmp_path / "base", tmp_path / "ckpt", tmp_path / "clip"
    for d in (base, ckpt, clip):
        d.mkdir()
    g = torch.Generator().manual_seed(0)
    q = torch.randn(8, 8, generator=g)
    save_file({"model.layers.0.self_attn.q_proj.weight": q, "model.embed_tokens.weight": torch.randn(10, 8, generator=g)},
              str(base / "model-00001-of-00002.safetensors"))
    save_file({"lm_head.weight": torch.randn(10, 8, generator=g)}, str(base / "model-00002-of-00002.safetensors"))
    torch.save({"model.mm_projector.norm.weight": torch.ones(8)}, str(ckpt / "mm_projector.bin"))
    save_file({"vision_model.pre_layrnorm.weight": torch.ones(4), "text_model.x": torch.zeros(1)}, str(clip / "model.safetensors"))
    got = dict(W.resize_vocab(W.iter_reference_checkpoint(str(ckpt), str(base), str(clip)), 11))
    assert set(got) == {"model.layers.0.self_attn.q_proj.weight", "model.embed_tokens.weight", "lm_head.weight",
                        "model.mm_projector.norm.weight", "model.vision_tower.vision_tower.vision_model.pre_layrnorm.weight"}
    assert got["model.embed_tokens.weight"].shape == (11, 8)
    assert torch.allclose(got["lm_head.weight"][10], got["lm_head.weight"][:10].mean(0))
    # LoRA: W + B @ A * alpha / r
    A, Bm = torch.randn(2, 8, generator=g), torch.randn(8, 2, generator=g)
    save_file({"base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight": A,
               "base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight": Bm}, str(ckpt / "adapter_model.safetensors"))
    json.dump({"r": 2, "lora_alpha": 4}, open(ckpt / "adapter_config.json", "w"))
    torch.save({"base_model.model.model.mm_projector.norm.weight": torch.full((8,), 2.0)}, str(ckpt / "non_lora_trainables.bin"))
    got = dict(W.iter_reference_checkpoint(str(ckpt), str(base), str(clip), lora=True))
    assert torch.allclose(got["model.layers.0.self_attn.q_proj.weight"], q + (Bm @ A) * 2.0, atol=1e-6)
    assert got["model.mm_projector.norm.weight"][0].item() == 2.0


def test_loader_rejects_what_the_engine_cannot_do():
    from vis_zephyr.model.builder import load_pretrained_model
    with pytest.raises(ValueError):
        load_pretrained_model("x", None, "llava-7b")


def test_nf4_quantiser_restatements_agree():
    """`load_4bit` (ref:vis_zephyr/model/builder.py:35-43 -> bitsandbytes nf4, blocksize 64): the engine-side quantiser (vz_hip/quant.py,
    torch.bucketize on the level midpoints) and the oracle's (plain loop over the 16 levels) are two restatements of the published algorithm -
    equal bit for bit on random weights, on the levels themselves, on ties and on an all-zero block.  Parity against bitsandbytes itself is
    unpinned (not vendored in the reference, not installed here); its double quantisation of the block scales is not restated."""
    import os
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import vz_oracle as O
    from vz_hip import quant
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(48, 512, generator=g) * 0.03).bfloat16().float()
    a, b = quant.fake_quantize_nf4(w), O.fake_quantize_nf4(w)
    assert torch.equal(a, b)
    rel = float((a - w).norm() / w.norm())
    assert 0.05 < rel < 0.13                                  # 4-bit normal-float on Gaussian weights: ~9 %
    lv = torch.tensor([list(quant.NF4_LEVELS) * 4]) * 0.37   # one block per 64: absmax 0.37, every element ON a level
    codes, absmax = quant.nf4_quantize(lv)
    assert codes[0, :16].tolist() == list(range(16)) and torch.allclose(absmax, torch.full_like(absmax, 0.37))
    assert torch.allclose(quant.nf4_dequantize(codes, absmax), lv, rtol=0, atol=1e-7) and torch.equal(O.fake_quantize_nf4(lv), quant.fake_quantize_nf4(lv))
    mid = torch.zeros(1, 64)
    mid[0, 0] = 1.0
    mid[0, 1] = (quant.NF4_LEVELS[9] + quant.NF4_LEVELS[10]) / 2          # a tie goes to the lower level in both
    assert torch.equal(quant.fake_quantize_nf4(mid), O.fake_quantize_nf4(mid))
    z = torch.zeros(2, 128)
    assert torch.equal(quant.fake_quantize_nf4(z), z) and torch.equal(O.fake_quantize_nf4(z), z)
    # rows are independent and blocks never straddle them: stacking q / k / v before quantising changes nothing
    q, k = w[:16], w[16:48]
    assert torch.equal(quant.fake_quantize_nf4(torch.cat([q, k])), torch.cat([quant.fake_quantize_nf4(q), quant.fake_quantize_nf4(k)]))


def test_prepare_inputs_for_generation_reattaches_images():
    """a4 (ref:vis_zephyr/model/language_model/vis_zephyr.py:144-170): pass-through that pops `images` / `images_size`, and puts
    them back only when they were given."""
    from vis_zephyr.model import VisZephyrForCausalLM
    f = VisZephyrForCausalLM.prepare_inputs_for_generation
    ids = torch.tensor([[1, 2, 3]])
    img = [torch.zeros(1, 3, 336, 336)]
    out = f(None, ids, past_key_values="pkv", inputs_embeds=None, attention_mask="m", images=img, images_size=[(1, 2)])
    assert out["input_ids"] is ids and out["past_key_values"] == "pkv" and out["attention_mask"] == "m"
    assert out["images"] is img and out["images_size"] == [(1, 2)]
    out = f(None, ids, use_cache=True)
    assert "images" not in out and "images_size" not in out and out["use_cache"] is True and out["inputs_embeds"] is None


def test_auto_registration_and_hub_cache_resolution(tmp_path, monkeypatch):
    """ref:vis_zephyr/model/language_model/vis_zephyr.py:173-174 registers the config / model with HF's Auto classes; hub ids
    resolve through the local cache only."""
    import json
    from transformers import AutoConfig, AutoModelForCausalLM
    from vis_zephyr.model import VisZephyrConfig, VisZephyrForCausalLM
    from vz_hip import weights as W
    d = tmp_path / "ckpt"
    d.mkdir()
    json.dump({"model_type": "vis_zephyr", "hidden_size": 4096, "mm_vision_tower": "openai/clip-vit-large-patch14-336"}, open(d / "config.json", "w"))
    c = AutoConfig.from_pretrained(str(d))
    assert type(c) is VisZephyrConfig and c.model_type == "vis_zephyr" and c.mm_vision_tower == "openai/clip-vit-large-patch14-336"
    assert AutoModelForCausalLM._model_mapping[VisZephyrConfig] is VisZephyrForCausalLM
    snap = tmp_path / "hub" / "models--openai--clip-vit-large-patch14-336" / "snapshots" / "abc"
    snap.mkdir(parents=True)
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hub"))
    import huggingface_hub.constants as hc
    monkeypatch.setattr(hc, "HF_HUB_CACHE", str(tmp_path / "hub"), raising=False)
    assert W.resolve_hub_path("openai/clip-vit-large-patch14-336") == str(snap)
    assert W.resolve_hub_path(str(d)) == str(d)
    import pytest
    with pytest.raises(FileNotFoundError):
        W.resolve_hub_path("HuggingFaceH4/zephyr-7b-beta")
