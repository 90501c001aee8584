#!/usr/bin/env python
"""Pin oracle/vip_oracle.py against Pillow itself and write tests/golden/vip_point.npz.  TEST INFRASTRUCTURE.

The reference's `image_blending(shape="point")` (`ref:vis_zephyr/model/vip_processor/conversation_generator.py:14-36,143-153,
170-175`, `ref:vis_zephyr/model/vip_processor/shape_draw.py:130-138`) is three Pillow calls on top of host-side geometry; its
module cannot be imported in this container (`shapely` absent), so the fixtures are produced by issuing
exactly those calls - `Image.new("RGBA")`, `ImageDraw.Draw(...).ellipse(box, fill=rgba, outline=rgba)`,
`Image.alpha_composite(image.convert("RGBA"), canvas).convert("RGB")` - on seeded images, boxes and colours.  Inputs are
regenerated from the seed by the tests; outputs are stored whole for the small cases and as CRC32 + a crop for the VCR-sized one.

    python oracle/pin_vip.py
"""
import os
import sys
import zlib

import numpy as np
from PIL import Image, ImageDraw

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import vip_oracle as V  # noqa: E402


def case(seed: int, w: int, h: int, n: int, edge: bool = False):
    """seeded inputs: image, boxes (floats, as draw_point computes them), colours."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    boxes, cols = [], []
    for i in range(n):
        if edge:        # centres on / beyond the border, degenerate and stretched boxes
            cx, cy = rng.uniform(-6, w + 6), rng.uniform(-6, h + 6)
            box = V.point_box(cx, cy, w, h, radius=float(rng.uniform(0.2, 30)), aspect_ratio=float(rng.uniform(0.5, 2.0)))
        else:
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            box = V.point_box(cx, cy, w, h)
        boxes.append(tuple(float(v) for v in box))
        cols.append(tuple(int(v) for v in rng.integers(0, 256, 3)) + (int(rng.integers(96, 256)),))
    return img, boxes, cols


def pillow_overlay(img, boxes, cols):
    im = Image.fromarray(img, "RGB")
    for box, col in zip(boxes, cols):
        im = im.convert("RGB")
        w, h = im.size
        vip = Image.new("RGBA", (w, h), (0, 0, 0, 0))
        ImageDraw.Draw(vip).ellipse(box, fill=col, outline=col)
        im = Image.alpha_composite(im.convert("RGBA"), vip).convert("RGB")
    return np.asarray(im)


CASES = [dict(seed=1, w=96, h=64, n=3), dict(seed=2, w=120, h=90, n=5, edge=True), dict(seed=3, w=336, h=336, n=2),
         dict(seed=4, w=1920, h=804, n=4), dict(seed=5, w=61, h=47, n=6, edge=True)]


def main():
    fx = {}
    for c in CASES:
        img, boxes, cols = case(**c)
        ref = pillow_overlay(img, boxes, cols)
        mine = V.point_overlay(img, boxes, cols)
        assert np.array_equal(ref, mine), f"oracle deviates from Pillow on case {c}"
        key = f"s{c['seed']}"
        fx[key + ".crc"] = np.uint32(zlib.crc32(ref.tobytes()))
        fx[key + ".changed"] = np.int64((ref != img).any(-1).sum())
        if ref.size <= 400000:
            fx[key + ".out"] = ref
        else:
            x0, y0 = int(boxes[0][0]) - 4, int(boxes[0][1]) - 4
            x0, y0 = max(0, min(x0, c["w"] - 128)), max(0, min(y0, c["h"] - 128))
            fx[key + ".crop_xy"] = np.array([x0, y0])
            fx[key + ".crop"] = ref[y0:y0 + 128, x0:x0 + 128]
        print(f"[pin_vip] case {c}: oracle == Pillow {Image.__version__ if hasattr(Image, '__version__') else ''}, "
              f"{int(fx[key + '.changed'])} pixels touched")
    # every (alpha, destination byte) pair of the compositing arithmetic, three source colours
    dst = np.tile(np.arange(256, dtype=np.uint8)[None, :, None], (1, 1, 3))
    table = np.zeros((256, 3, 256, 3), dtype=np.uint8)
    for sa in range(256):
        for j, col in enumerate([(255, 0, 0), (17, 99, 203), (255, 255, 255)]):
            ov = Image.new("RGBA", (256, 1), col + (sa,))
            ref = np.asarray(Image.alpha_composite(Image.fromarray(dst, "RGB").convert("RGBA"), ov).convert("RGB"))[0]
            assert np.array_equal(ref, V.composite_over_opaque(dst, col + (sa,))[0])
            table[sa, j] = ref
    fx["composite.crc"] = np.uint32(zlib.crc32(table.tobytes()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "vip_point.npz"), **fx)
    print("[pin_vip] OK - fixtures written")


if __name__ == "__main__":
    main()
