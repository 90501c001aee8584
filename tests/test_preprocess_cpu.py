"""Anyres preprocessing: the numpy oracle against the golden tiles captured from the reference's own function
(oracle/pin_preprocess.py), against Pillow, and the host mirror's PIL path against the oracle."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
from pin_preprocess import CASES, PINPOINTS, synth_image  # noqa: E402  (seeded inputs + the candidate grid; no reference import)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(REPO, "tests", "golden", "preprocess.npz"))


@pytest.mark.parametrize("name,h,w,seed", CASES)
def test_oracle_reproduces_reference_tiles(gold, name, h, w, seed):
    from oracle import preprocess_oracle as P
    out = P.process_any_resolution_image(synth_image(h, w, seed), PINPOINTS)
    assert tuple(out.shape) == tuple(gold[name + ".shape"].tolist())
    assert np.array_equal(out[:, :, ::7, ::5], gold[name + ".sub"])                     # every float, exactly
    assert abs(float(out.astype(np.float64).sum()) - float(gold[name + ".sum"])) <= 1e-9 * float(gold[name + ".abssum"])


def test_oracle_resize_is_pillow_lanczos():
    from PIL import Image
    from oracle import preprocess_oracle as P
    rng = np.random.default_rng(7)
    for (h, w), size in [((97, 131), (336, 336)), ((700, 500), (240, 336)), ((336, 400), (336, 336)), ((64, 64), (64, 64))]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize(size, Image.Resampling.LANCZOS))
        assert np.array_equal(P.resize_lanczos(img, size), ref)


def test_product_coefficients_equal_oracle():
    """vz_hip/preprocess.py carries its own copy of Pillow's weight computation (the product may not import oracle/)."""
    from oracle import preprocess_oracle as P
    from vz_hip import preprocess as D
    for a, b in [(1920, 1008), (804, 422), (640, 336), (61, 336), (500, 672), (336, 337)]:
        b0, c0, k0 = P.precompute_coeffs(a, b)
        b1, c1, k1 = D.lanczos_coeffs(a, b)
        assert k0 == k1 and np.array_equal(b0, b1) and np.array_equal(c0, c1)
    for size in [(640, 480), (1920, 804), (400, 500), (61, 90), (336, 336)]:
        assert D.select_best_fit_resolution(size, PINPOINTS) == tuple(P.select_best_fit_resolution(size, PINPOINTS))


def test_mirror_pil_path_equals_oracle():
    from PIL import Image
    from oracle import preprocess_oracle as P
    from vis_zephyr.model.multi_scale_process import process_any_resolution_image
    from vis_zephyr.model.vision_encoder.vision_encoder import _make_image_processor
    proc = _make_image_processor("openai/clip-vit-large-patch14-336", 336)
    if proc is None:
        pytest.skip("transformers image processor unavailable")
    img = synth_image(300, 777, 11)
    got = process_any_resolution_image(Image.fromarray(img), proc, str(PINPOINTS)).numpy()
    assert np.array_equal(got, P.process_any_resolution_image(img, PINPOINTS))
