"""CPU restatement of the ViP "point" overlay (second half).  TEST INFRASTRUCTURE ONLY.

What the reference does for the `vcr_qa` / `vcr_qar` visual-prompt styles (`ref:vis_zephyr/model/vip_processor/configuration.py:18-19`
= shape "point", style "constant"): `image_blending` (`ref:vis_zephyr/model/vip_processor/conversation_generator.py:14-36,143-153,
170-175`) draws ONE filled ellipse per instance on a transparent RGBA canvas (`draw_point`,
`ref:vis_zephyr/model/vip_processor/shape_draw.py:100-138`: `ImageDraw.ellipse(bbox, fill=rgba, outline=rgba)`; centre drawn by a
host-side Gaussian, radius = max(int(8 * max(w, h) / 336), 1), aspect 1) and composites it: `Image.alpha_composite(image.convert
("RGBA"), canvas).convert("RGB")`.

The pixel arithmetic is third-party - Pillow (`ref:environment.yaml` pins pillow; installed here: 12.2.0), integer throughout:
  * `ImageDraw.ellipse` with outline == fill draws the fill pass only; `_draw_ellipse` truncates the float box to ints and
    `ellipseNew` (src/libImaging/Draw.c) walks a quarter ellipse in DOUBLED coordinates with an error-minimising step
    (`quarter_next`), emitting one `hline` per scanline: x0 + (X + a) / 2, y0 + (Y + b) / 2;
  * `ImagingAlphaComposite` (src/libImaging/AlphaComposite.c): 7 extra bits of precision, `SHIFTFORDIV255` rounding.
Restated below in plain Python / numpy.  PINNED against Pillow itself on seeded boxes (`oracle/pin_vip.py` ->
`tests/golden/vip_point.npz`, re-checked by tests/test_vip_cpu.py wherever Pillow imports); the reference's own module cannot
be imported here (`shapely` is absent: SURVEY.md section 8c) and holds no arithmetic beyond the calls above.
"""
from __future__ import annotations

import numpy as np


def _quarter_points(a: int, b: int):
    """Pillow's quarter_state walk: points (cx, cy) of the quarter ellipse with doubled axes a, b, from (a, b % 2) to (a % 2, b)."""
    if a < 0 or b < 0:
        return
    a2, b2 = a * a, b * b
    a2b2 = a2 * b2
    cx, cy, ex, ey = a, b % 2, a % 2, b

    def delta(x, y):
        return abs(a2 * y * y + b2 * x * x - a2b2)

    while True:
        yield cx, cy
        if cx == ex and cy == ey:
            return
        nx, ny = cx, cy + 2
        nd = delta(nx, ny)
        if nx > 1:
            d = delta(cx - 2, cy + 2)
            if nd > d:
                nx, ny, nd = cx - 2, cy + 2, d
            d = delta(cx - 2, cy)
            if nd > d:
                nx, ny = cx - 2, cy
        cx, cy = nx, ny


def filled_ellipse_spans(x0: int, y0: int, x1: int, y1: int):
    """[(y, xl, xr)] inclusive pixel spans of `ImageDraw.ellipse((x0, y0, x1, y1), fill=...)` before clipping to the image."""
    a, b = x1 - x0, y1 - y0
    if a < 0 or b < 0:
        return []
    # per doubled scanline Y >= 0 the outer quarter's LAST point with that cy gives the right end r; the fill reaches from -r to r
    width = {}
    for cx, cy in _quarter_points(a, b):
        if cy not in width:
            width[cy] = cx          # ellipse_next takes the FIRST point of a scanline (pr / py are loaded before the skip loop)
    spans = []
    for Y, r in width.items():
        for yy in ({Y, -Y}):
            spans.append((y0 + (yy + b) // 2, x0 + (-r + a) // 2, x0 + (r + a) // 2))
    return spans


def draw_point_mask(h: int, w: int, box) -> np.ndarray:
    """bool [h, w]: pixels `ImageDraw.Draw(RGBA canvas).ellipse(box, fill=c, outline=c)` sets (box = 4 floats, truncated like C's (int))."""
    x0, y0, x1, y1 = (int(v) for v in box)          # (int) of a double truncates toward zero
    m = np.zeros((h, w), dtype=bool)
    for y, xl, xr in filled_ellipse_spans(x0, y0, x1, y1):
        if 0 <= y < h:
            xl, xr = max(xl, 0), min(xr, w - 1)
            if xl <= xr:
                m[y, xl:xr + 1] = True
    return m


def composite_over_opaque(dst_rgb: np.ndarray, src_rgba) -> np.ndarray:
    """ImagingAlphaComposite(dst = opaque RGB, src = constant RGBA) followed by convert("RGB"): uint8 [...,3]."""
    sr, sg, sb, sa = (int(v) for v in src_rgba)
    d = dst_rgb.astype(np.uint32)
    if sa == 0:
        return dst_rgb.copy()
    blend = 255 * (255 - sa)
    outa255 = sa * 255 + blend
    coef1 = sa * 255 * 255 * 128 // outa255
    coef2 = 255 * 128 - coef1
    src = np.array([sr, sg, sb], dtype=np.uint32)
    t = src * np.uint32(coef1) + d * np.uint32(coef2) + np.uint32(0x80 << 7)
    t = ((t >> 8) + t) >> 8
    return (t >> 7).astype(np.uint8)


def point_overlay(image_rgb: np.ndarray, boxes, colors) -> np.ndarray:
    """image uint8 [h, w, 3]; one ellipse box (x0, y0, x1, y1 floats) and one RGBA colour per instance, applied in order
    (each `image_blending` call composites onto the result of the previous one)."""
    out = image_rgb.copy()
    h, w = out.shape[:2]
    for box, rgba in zip(boxes, colors):
        m = draw_point_mask(h, w, box)
        out[m] = composite_over_opaque(out[m], rgba)
    return out


def point_box(cx: float, cy: float, img_w: int, img_h: int, image_size_anchor: int = 336, aspect_ratio: float = 1.0, radius=None):
    """the ellipse box `draw_point` hands to Pillow for the 'constant' style (conversation_generator.py:143-146, shape_draw.py:130-134)."""
    if radius is None:
        radius = max(int(8 * max(img_w, img_h) / image_size_anchor), 1)
    xr, yr = radius * aspect_ratio, radius / aspect_ratio
    return (cx - xr, cy - yr, cx + xr, cy + yr)
