"""The ViP "point" overlay on the device (csrc/preprocess.hip vip_point_kernel) - bit-exact against oracle/vip_oracle.py and
the Pillow fixtures, then through the device anyres pipeline (overlay -> tiles) as the VCR eval feeds it."""
import os
import zlib

import numpy as np
import pytest
import torch

from oracle import pin_vip, vip_oracle as V
from util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("c", pin_vip.CASES, ids=lambda c: f"s{c['seed']}")
def test_device_overlay_is_bit_exact(c):
    from vz_hip import preprocess as P
    g = np.load(os.path.join(GOLDEN, "vip_point.npz"))
    img, boxes, cols = pin_vip.case(**c)
    got = P.vip_point_overlay(torch.from_numpy(img).cuda().contiguous(), boxes, cols).cpu().numpy()
    assert np.array_equal(got, V.point_overlay(img, boxes, cols))
    assert zlib.crc32(got.tobytes()) == int(g[f"s{c['seed']}.crc"])


def test_device_overlay_fuzz_and_limits():
    from vz_hip import preprocess as P
    for seed in range(200, 260):
        w, h = 33 + (seed * 7) % 300, 21 + (seed * 13) % 200
        img, boxes, cols = pin_vip.case(seed, w, h, 1 + seed % 4, edge=bool(seed % 3))
        got = P.vip_point_overlay(torch.from_numpy(img).cuda().contiguous(), boxes, cols).cpu().numpy()
        assert np.array_equal(got, V.point_overlay(img, boxes, cols)), seed
    # alpha 0 and inverted boxes leave the image alone; an ellipse larger than the image covers it
    img = np.random.default_rng(0).integers(0, 256, (50, 70, 3), dtype=np.uint8)
    t = torch.from_numpy(img).cuda()
    assert np.array_equal(P.vip_point_overlay(t.clone(), [(5, 5, 30, 30)], [(1, 2, 3, 0)]).cpu().numpy(), img)
    assert np.array_equal(P.vip_point_overlay(t.clone(), [(30, 30, 5, 5)], [(1, 2, 3, 200)]).cpu().numpy(), img)
    big = [(-200.0, -200.0, 300.0, 300.0)]
    assert np.array_equal(P.vip_point_overlay(t.clone(), big, [(9, 8, 7, 255)]).cpu().numpy(), V.point_overlay(img, big, [(9, 8, 7, 255)]))
    with pytest.raises(ValueError):
        P.vip_point_overlay(t.clone(), [(0.0, 0.0, 10.0, 9000.0)], [(1, 2, 3, 200)])


def test_overlay_then_anyres_tiles_matches_host_pipeline():
    """VCR item shape: 1920 x 804 frame, points drawn on the device, then the device anyres pipeline -> the tiles of the same
    frame overlaid by the oracle on the host."""
    from oracle import preprocess_oracle as PO
    from vz_hip import preprocess as P
    pin = [(336, 672), (672, 336), (336, 1008), (1008, 336)]
    img, boxes, cols = pin_vip.case(4, 1920, 804, 4)
    pre = P.AnyresPreprocessor("cuda:0")
    dev = P.vip_point_overlay(torch.from_numpy(img).cuda().contiguous(), boxes, cols)
    tiles = pre(dev, str([list(p) for p in pin]))
    ref = PO.process_any_resolution_image(V.point_overlay(img, boxes, cols), pin)
    ref = torch.from_numpy(np.asarray(ref)).to(torch.bfloat16)
    assert tuple(tiles.shape) == tuple(ref.shape) == (4, 3, 336, 336)
    assert torch.equal(tiles.cpu(), ref)
