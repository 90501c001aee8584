#!/usr/bin/env python
"""Pin oracle/preprocess_oracle.py against the REFERENCE's own anyres preprocessing and emit golden tiles.

TEST INFRASTRUCTURE, build container only (/root/reference present): imports the reference's
`vis_zephyr.model.multi_scale_process.process_any_resolution_image` (PIL LANCZOS + HF CLIPImageProcessor), runs it on
seeded synthetic 8-bit images, requires the numpy oracle to reproduce every output float EXACTLY, and writes subsampled
tiles + checksums to tests/golden/preprocess.npz (inputs regenerate from seeds).

    python oracle/pin_preprocess.py"""
import importlib.util
import os
import sys

import numpy as np
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import preprocess_oracle as P  # noqa: E402

PINPOINTS = [[336, 672], [672, 336], [336, 1008], [1008, 336], [672, 672]]
CASES = [("vga", 480, 640, 1), ("vcr", 804, 1920, 2), ("portrait", 500, 400, 3), ("tiny", 90, 61, 4), ("exact", 336, 336, 5)]


def synth_image(h, w, seed):
    """noise + smooth gradients + a hard edge: exercises ringing, clipping at 0 / 255 and the letterbox border."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    img[: h // 3] = np.stack([(xx[: h // 3] * 255 // max(1, w - 1)), (yy[: h // 3] * 255 // max(1, h - 1)), 255 - (xx[: h // 3] * 255 // max(1, w - 1))], -1).astype(np.uint8)
    img[h // 3: h // 2, : w // 2] = 0
    img[h // 3: h // 2, w // 2:] = 255
    return img


def main():
    spec = importlib.util.spec_from_file_location("ref_msp", "/root/reference/vis_zephyr/model/multi_scale_process.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, resample=3,
                              image_mean=list(P.CLIP_MEAN), image_std=list(P.CLIP_STD), do_convert_rgb=True)
    fx = {}
    for name, h, w, seed in CASES:
        img = synth_image(h, w, seed)
        out_ref = ref.process_any_resolution_image(Image.fromarray(img), proc, str(PINPOINTS)).numpy()
        mine = P.process_any_resolution_image(img, PINPOINTS)
        assert out_ref.shape == mine.shape, (name, out_ref.shape, mine.shape)
        exact = np.array_equal(out_ref, mine)
        print(f"[pin] {name:9s} {h}x{w} -> tiles {out_ref.shape}  oracle == reference: {exact}  max|diff| {np.abs(out_ref - mine).max():.3e}", flush=True)
        assert exact, name
        fx[name + ".shape"] = np.array(out_ref.shape, dtype=np.int64)
        fx[name + ".sub"] = out_ref[:, :, ::7, ::5].astype(np.float32)
        fx[name + ".sum"] = np.float64(out_ref.astype(np.float64).sum())
        fx[name + ".abssum"] = np.float64(np.abs(out_ref.astype(np.float64)).sum())
        fx[name + ".hws"] = np.array([h, w, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "preprocess.npz"), **fx)
    print("[pin] OK - fixtures written")


if __name__ == "__main__":
    main()
