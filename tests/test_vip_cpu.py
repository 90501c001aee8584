"""oracle/vip_oracle.py against the committed Pillow fixtures (tests/golden/vip_point.npz, oracle/pin_vip.py) and, where Pillow
imports, against Pillow itself on fresh seeded boxes."""
import os
import zlib

import numpy as np
import pytest

from oracle import pin_vip, vip_oracle as V
from util import GOLDEN


def _gold():
    z = np.load(os.path.join(GOLDEN, "vip_point.npz"))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("c", pin_vip.CASES, ids=lambda c: f"s{c['seed']}")
def test_oracle_reproduces_pillow_fixtures(c):
    g = _gold()
    img, boxes, cols = pin_vip.case(**c)
    out = V.point_overlay(img, boxes, cols)
    key = f"s{c['seed']}"
    assert zlib.crc32(out.tobytes()) == int(g[key + ".crc"])
    assert int((out != img).any(-1).sum()) == int(g[key + ".changed"])
    if key + ".out" in g:
        assert np.array_equal(out, g[key + ".out"])
    else:
        x0, y0 = g[key + ".crop_xy"]
        assert np.array_equal(out[y0:y0 + 128, x0:x0 + 128], g[key + ".crop"])


def test_composite_table_and_live_pillow():
    g = _gold()
    dst = np.tile(np.arange(256, dtype=np.uint8)[None, :, None], (1, 1, 3))
    table = np.zeros((256, 3, 256, 3), dtype=np.uint8)
    for sa in range(256):
        for j, col in enumerate([(255, 0, 0), (17, 99, 203), (255, 255, 255)]):
            table[sa, j] = V.composite_over_opaque(dst, col + (sa,))[0]
    assert zlib.crc32(table.tobytes()) == int(g["composite.crc"])
    pytest.importorskip("PIL")
    for seed in range(100, 140):
        img, boxes, cols = pin_vip.case(seed, 40 + seed % 50, 30 + seed % 37, 3, edge=bool(seed % 2))
        assert np.array_equal(V.point_overlay(img, boxes, cols), pin_vip.pillow_overlay(img, boxes, cols)), seed
