"""Device-side anyres preprocessing (csrc/preprocess.hip) against the CPU oracle and the golden tiles of the reference:
integer / table work, so the bar is BIT-EXACT (the bf16 cast of the reference's float32 output)."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
from pin_preprocess import CASES, PINPOINTS, synth_image  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pre():
    from vz_hip.preprocess import AnyresPreprocessor
    return AnyresPreprocessor("cuda:0")


@pytest.mark.parametrize("hw,size", [((97, 131), (336, 336)), ((804, 1920), (1008, 422)), ((700, 500), (240, 336)),
                                     ((336, 400), (336, 336)), ((400, 336), (336, 336)), ((64, 64), (64, 64)), ((3, 5), (336, 336)),
                                     ((2000, 3000), (672, 448))])
def test_resize_is_bit_exact(pre, hw, size):
    from oracle import preprocess_oracle as P
    rng = np.random.default_rng(hw[0] * 7 + hw[1])
    img = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    img[: hw[0] // 2, : hw[1] // 2] = 255          # saturated block next to noise: ringing must clip exactly as Pillow clips
    got = pre.resize_lanczos(torch.from_numpy(img).cuda(), size).cpu().numpy()
    ref = P.resize_lanczos(img, size)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} bytes differ, max {int(np.abs(got.astype(int) - ref.astype(int)).max())}"


@pytest.mark.parametrize("name,h,w,seed", CASES)
def test_tiles_match_oracle_and_reference_golden(pre, name, h, w, seed):
    from oracle import preprocess_oracle as P
    img = synth_image(h, w, seed)
    got = pre(torch.from_numpy(img), PINPOINTS)
    ref32 = P.process_any_resolution_image(img, PINPOINTS)
    assert got.dtype == torch.bfloat16 and tuple(got.shape) == ref32.shape
    ref = torch.from_numpy(ref32).bfloat16()
    assert torch.equal(got.cpu(), ref), f"{int((got.cpu() != ref).sum())} elements differ"
    gold = np.load(os.path.join(REPO, "tests", "golden", "preprocess.npz"))
    assert torch.equal(got.cpu()[:, :, ::7, ::5], torch.from_numpy(gold[name + ".sub"]).bfloat16())     # the reference's own output


def test_mirror_device_entry_point_and_engine_roundtrip(pre):
    """`process_any_resolution_image_device` (drop-in next to the PIL function) feeds the vision tower directly."""
    from PIL import Image
    from vis_zephyr.model.multi_scale_process import process_any_resolution_image, process_any_resolution_image_device
    from vis_zephyr.model.vision_encoder.vision_encoder import _make_image_processor
    proc = _make_image_processor("openai/clip-vit-large-patch14-336", 336)
    if proc is None:
        pytest.skip("transformers image processor unavailable")
    img = Image.fromarray(synth_image(611, 905, 21))
    host = process_any_resolution_image(img, proc, str(PINPOINTS))
    dev = process_any_resolution_image_device(img, proc, str(PINPOINTS))
    assert torch.equal(dev.cpu(), host.bfloat16())
    with pytest.raises(ValueError):
        pre.resize_lanczos(torch.zeros(4, 4, 3, dtype=torch.uint8, device="cuda"), (0, 5))
