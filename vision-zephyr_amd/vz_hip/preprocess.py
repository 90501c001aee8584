"""Device-side anyres preprocessing: the reference's `process_any_resolution_image`
(ref:vis_zephyr/model/multi_scale_process.py:133-171) with the pixel work on the GPU.

Host: geometry (best-fit grid, letterbox offsets: integer / float arithmetic of the reference, a few operations per
image) and the resampling weights - Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c) for
LANCZOS, which depend on (input size, output size) only and are cached per size pair on the device.
Device (csrc/preprocess.hip): the two 8-bit resampling passes, the letterbox + tiling, CLIP rescale / normalise through a
256-entry table per channel.  Output: bf16 [1 + n_crops, 3, 336, 336], bit-identical to bf16(reference output).
No CPU fallback: without the HIP library this module raises."""
from __future__ import annotations

import ast
import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import binding as B

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    return _sinc(x) * _sinc(x / 3) if -3.0 <= x < 3.0 else 0.0


def lanczos_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out, 2], coefs int32 [out, ksize], ksize): Pillow's weights for resizing `in_size` -> `out_size`."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        for x, w in enumerate(k):
            if ww != 0.0:
                w = w / ww
            coefs[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
        bounds[xx] = (xmin, xmax)
    return bounds, coefs, ksize


def select_best_fit_resolution(original: Tuple[int, int], candidates: Sequence[Sequence[int]]) -> Tuple[int, int]:
    ow, oh = original
    best, best_key = None, None
    for w, h in candidates:
        s = min(w / ow, h / oh)
        kept = min(int(ow * s) * int(oh * s), ow * oh)
        key = (kept, -(w * h - kept))
        if best_key is None or key > best_key:
            best, best_key = (int(w), int(h)), key
    return best


class AnyresPreprocessor:
    """`pre = AnyresPreprocessor("cuda:0"); tiles = pre(image_u8_hwc, grid_pinpoints)`"""

    def __init__(self, device="cuda:0", side: int = 336, mean=CLIP_MEAN, std=CLIP_STD):
        B.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("vz_hip.preprocess needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.side = side
        # CLIPImageProcessor: rescale = float32(float64(u8) * (1/255)), normalise = (x - mean) / std in float32
        v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
        lut = (v[None, :] - np.array(mean, dtype=np.float32)[:, None]) / np.array(std, dtype=np.float32)[:, None]
        self.lut = torch.from_numpy(lut.astype(np.float32)).to(self.device).to(torch.bfloat16).contiguous()
        self._coef: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}

    def _coeffs(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        if key not in self._coef:
            b, c, k = lanczos_coeffs(in_size, out_size)
            self._coef[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(c).to(self.device), k)
        return self._coef[key]

    def resize_lanczos(self, img: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
        """uint8 [H, W, 3] on the device -> uint8 [h2, w2, 3]; `size` = (w2, h2) as PIL takes it."""
        assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_cuda and img.is_contiguous()
        w2, h2 = int(size[0]), int(size[1])
        h, w = int(img.shape[0]), int(img.shape[1])
        if w2 <= 0 or h2 <= 0:
            raise ValueError("height and width must be > 0")          # as PIL's Image.resize
        dst = torch.empty(h2, w2, 3, dtype=torch.uint8, device=self.device)
        xb = xc = yb = yc = None
        kx = ky = 0
        if w2 != w:
            xb, xc, kx = self._coeffs(w, w2)
        if h2 != h:
            yb, yc, ky = self._coeffs(h, h2)
        tmp = torch.empty(h, w2, 3, dtype=torch.uint8, device=self.device) if (w2 != w and h2 != h) else None
        B.check(B.lib().vz_op_resample_u8(B.ptr(img), h, w, B.ptr(tmp), B.ptr(dst), h2, w2, B.ptr(xb), B.ptr(xc), kx, B.ptr(yb), B.ptr(yc),
                                          ky, B.stream_ptr(self.device)))
        return dst

    def __call__(self, image, grid_pinpoints) -> torch.Tensor:
        """image: uint8 [H, W, 3] tensor (any device), numpy array or PIL image (RGB) -> bf16 [1 + n_crops, 3, side, side]."""
        if not isinstance(image, torch.Tensor):
            arr = np.asarray(image.convert("RGB") if hasattr(image, "convert") else image)
            image = torch.from_numpy(np.array(arr, copy=True))
        img = image.to(self.device).contiguous()
        res = grid_pinpoints
        while isinstance(res, str):
            res = ast.literal_eval(res)
        h, w = int(img.shape[0]), int(img.shape[1])
        side = self.side
        tw, th = select_best_fit_resolution((w, h), res)
        s = min(tw / w, th / h)
        nw, nh = int(w * s), int(h * s)
        px, py = (tw - nw) // 2, (th - nh) // 2
        resized = self.resize_lanczos(img, (nw, nh))
        glob = self.resize_lanczos(img, (side, side))
        gw, gh = tw // side, th // side
        out = torch.empty(1 + gw * gh, 3, side, side, dtype=torch.bfloat16, device=self.device)
        B.check(B.lib().vz_op_anyres_tiles(B.ptr(glob), B.ptr(resized), nh, nw, px, py, gw, gh, side, B.ptr(self.lut), B.ptr(out),
                                           B.stream_ptr(self.device)))
        return out


def vip_point_box(cx: float, cy: float, img_w: int, img_h: int, image_size_anchor: int = 336, aspect_ratio: float = 1.0, radius=None):
    """the ellipse box the reference's `draw_point` hands to Pillow ('constant' style of vcr_qa / vcr_qar:
    ref:vis_zephyr/model/vip_processor/conversation_generator.py:143-146, shape_draw.py:130-134)."""
    if radius is None:
        radius = max(int(8 * max(img_w, img_h) / image_size_anchor), 1)
    xr, yr = radius * aspect_ratio, radius / aspect_ratio
    return (cx - xr, cy - yr, cx + xr, cy + yr)


def vip_point_overlay(image: torch.Tensor, boxes, colors) -> torch.Tensor:
    """The pixel work of `image_blending(shape="point")` on the device: `image` uint8 [H, W, 3] (device tensor, modified in
    place and returned); one ellipse box (x0, y0, x1, y1 floats, as `draw_point` computes it) and one (r, g, b, a) per
    instance, composited in order.  The centre of each point is host-side geometry (the reference samples it with scipy's
    multivariate_normal around the instance box, shape_draw.py:106-128) and stays the caller's."""
    assert image.dtype == torch.uint8 and image.dim() == 3 and image.shape[2] == 3 and image.is_cuda and image.is_contiguous()
    h, w = int(image.shape[0]), int(image.shape[1])
    for box, (r, g, b, a) in zip(boxes, colors):
        x0, y0, x1, y1 = (int(v) for v in box)           # Pillow's _draw_ellipse: (int) of each double
        rgba = (int(r) & 255) | ((int(g) & 255) << 8) | ((int(b) & 255) << 16) | ((int(a) & 255) << 24)
        B.check(B.lib().vz_op_vip_point(B.ptr(image), h, w, x0, y0, x1, y1, rgba, B.stream_ptr(image.device)))
    return image
