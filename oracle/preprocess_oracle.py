"""CPU oracle of the anyres preprocessing.  TEST INFRASTRUCTURE: imported by tests/ only.

Restates, in numpy, what the reference's `process_any_resolution_image` does to an 8-bit RGB image
(ref:vis_zephyr/model/multi_scale_process.py:29-183) including the arithmetic that lives in third-party code:

  * `Image.resize(size, LANCZOS)` - Pillow (pinned 11.x by ref:environment.yaml; 12.2 installed) `src/libImaging/Resample.c`:
    `precompute_coeffs` (double-precision windowed-sinc weights per output pixel over support 3 * max(scale, 1), normalised),
    `normalize_coeffs_8bpc` (weights -> int32 with PRECISION_BITS = 32 - 8 - 2 = 22, round half away from zero),
    `ImagingResampleHorizontal_8bpc` then `ImagingResampleVertical_8bpc` (accumulator starts at 1 << 21, arithmetic shift
    by 22, clip to [0, 255]; the intermediate image is 8-bit).  A resize to the same size is a copy.
  * the letterbox paste on black, the row-major 336 x 336 crops, the prepended global view (ref :150-171);
  * `CLIPImageProcessor.preprocess` on a 336 x 336 tile: resize and centre crop are identities, rescale = float32(float64(u8)
    / 255), normalise = (x - mean) / std in float32 (hf:image_transforms.py rescale / normalize).

Pinned against Pillow itself and against the reference's function (tests/test_preprocess_cpu.py, golden tiles in
tests/golden/preprocess.npz made by oracle/pin_preprocess.py)."""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def precompute_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow `precompute_coeffs` + `normalize_coeffs_8bpc` for the full-image box and the LANCZOS filter.
    Returns (bounds int32 [out, 2] = (first input index, count), coefs int32 [out, ksize], ksize)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            coefs[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs, ksize


def _resample_axis(img: np.ndarray, bounds: np.ndarray, coefs: np.ndarray, axis: int) -> np.ndarray:
    """one 8-bit pass along `axis` (1 = horizontal, 0 = vertical) of an [H, W, 3] uint8 image."""
    src = img.astype(np.int64)
    out_len = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_len
    out = np.empty(shape, dtype=np.uint8)
    for i in range(out_len):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        k = coefs[i, :n].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + n, :] * k[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            out[:, i, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        else:
            acc = (src[lo:lo + n, :, :] * k[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            out[i, :, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


def resize_lanczos(img: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """`Image.fromarray(img).resize((w, h), LANCZOS)` on an [H, W, 3] uint8 array: horizontal pass, then vertical."""
    w2, h2 = size
    h, w = img.shape[:2]
    out = img
    if w2 != w:
        bx, cx, _ = precompute_coeffs(w, w2)
        out = _resample_axis(out, bx, cx, 1)
    if h2 != h:
        by, cy, _ = precompute_coeffs(h, h2)
        out = _resample_axis(out, by, cy, 0)
    return out.copy() if out is img else out


def select_best_fit_resolution(original: Tuple[int, int], candidates: Sequence[Tuple[int, int]]) -> Tuple[int, int]:
    """ref multi_scale_process.py:29-67."""
    ow, oh = original
    best, max_eff, min_waste = None, 0, float("inf")
    for w, h in candidates:
        s = min(w / ow, h / oh)
        dw, dh = int(ow * s), int(oh * s)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (w, h)
    return best


def letterbox_geometry(size: Tuple[int, int], target: Tuple[int, int]) -> Tuple[int, int, int, int]:
    """ref :70-94: (new_w, new_h, paste_x, paste_y)."""
    ow, oh = size
    tw, th = target
    s = min(tw / ow, th / oh)
    nw, nh = int(ow * s), int(oh * s)
    return nw, nh, (tw - nw) // 2, (th - nh) // 2


def normalize_lut() -> np.ndarray:
    """float32 [3, 256]: CLIPImageProcessor's rescale + normalize for every 8-bit value (hf:image_transforms.py)."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    mean = np.array(CLIP_MEAN, dtype=np.float32)
    std = np.array(CLIP_STD, dtype=np.float32)
    return ((v[None, :] - mean[:, None]) / std[:, None]).astype(np.float32)


def anyres_tiles_u8(img: np.ndarray, grid_pinpoints: Sequence[Tuple[int, int]], side: int = 336) -> np.ndarray:
    """uint8 [1 + n_crops, side, side, 3]: the global view, then the crops of the letterboxed image, row-major (ref :133-171)."""
    h, w = img.shape[:2]
    tw, th = select_best_fit_resolution((w, h), grid_pinpoints)
    nw, nh, px, py = letterbox_geometry((w, h), (tw, th))
    canvas = np.zeros((th, tw, 3), dtype=np.uint8)
    canvas[py:py + nh, px:px + nw] = resize_lanczos(img, (nw, nh))
    tiles: List[np.ndarray] = [resize_lanczos(img, (side, side))]
    for y in range(0, th, side):
        for x in range(0, tw, side):
            tiles.append(canvas[y:y + side, x:x + side])
    return np.stack(tiles)


def process_any_resolution_image(img: np.ndarray, grid_pinpoints: Sequence[Tuple[int, int]], side: int = 336) -> np.ndarray:
    """float32 [N, 3, side, side] = what the reference hands to the vision tower."""
    t = anyres_tiles_u8(img, grid_pinpoints, side)                   # [N, s, s, 3]
    lut = normalize_lut()
    out = np.empty((t.shape[0], 3, side, side), dtype=np.float32)
    for c in range(3):
        out[:, c] = lut[c][t[..., c]]
    return out
