"""Any-resolution tiling of an input image (API of ref:vis_zephyr/model/multi_scale_process.py:12-211).

Host-side PIL work feeding the path: pick the grid resolution that keeps the most image area,
letterbox the image into it (LANCZOS), cut 336x336 crops row-major, prepend the whole image resized to 336x336,
CLIP-normalise every tile -> [1 + n_crops, 3, 336, 336].
"""
from __future__ import annotations

import ast
from typing import List, Sequence, Tuple

import torch
from PIL import Image


def _robust_literal_eval(value):
    """`"'[[336, 672]]'"` (a literal quoted once or twice on the command line) -> the Python object."""
    while isinstance(value, str):
        try:
            value = ast.literal_eval(value)
        except (ValueError, SyntaxError):
            break
    return value


def select_best_fit_resolution(original_resolution: Tuple[int, int], possible_resolutions: Sequence[Tuple[int, int]]):
    """(w, h) of the candidate that preserves the largest image area after a fit-inside scale; ties go to the least
    padding, then to the first candidate."""
    ow, oh = original_resolution
    best, best_key = None, None
    for w, h in possible_resolutions:
        s = min(w / ow, h / oh)
        kept = min(int(ow * s) * int(oh * s), ow * oh)
        key = (kept, -(w * h - kept))
        if best_key is None or key > best_key:
            best, best_key = (w, h), key
    return best


def resize_pad_image(image: Image.Image, target_res: Tuple[int, int]) -> Image.Image:
    tw, th = target_res
    s = min(tw / image.width, th / image.height)
    nw, nh = int(image.width * s), int(image.height * s)
    canvas = Image.new("RGB", (tw, th), (0, 0, 0))
    canvas.paste(image.resize((nw, nh), Image.Resampling.LANCZOS), ((tw - nw) // 2, (th - nh) // 2))
    return canvas


def divide_to_patches(image: Image.Image, patch_size: int) -> List[Image.Image]:
    return [image.crop((x, y, x + patch_size, y + patch_size))
            for y in range(0, image.height, patch_size) for x in range(0, image.width, patch_size)]


def calculate_grid_shape(image_size: Tuple[int, int], grid_pinpoints, patch_size: int) -> Tuple[int, int]:
    res = _robust_literal_eval(grid_pinpoints)
    if not isinstance(res, list):
        raise ValueError(f"grid_pinpoints did not evaluate to a list: {grid_pinpoints}")
    w, h = select_best_fit_resolution(image_size, res)
    return w // patch_size, h // patch_size


def process_any_resolution_image(image: Image.Image, processor, grid_pinpoints) -> torch.Tensor:
    res = grid_pinpoints if isinstance(grid_pinpoints, list) else _robust_literal_eval(grid_pinpoints)
    side = processor.crop_size["height"]
    padded = resize_pad_image(image, select_best_fit_resolution(image.size, res))
    tiles = [image.resize((side, side), Image.Resampling.LANCZOS)] + divide_to_patches(padded, side)
    return torch.cat([processor.preprocess(t, return_tensors="pt")["pixel_values"] for t in tiles], dim=0)


_DEVICE_PRE = {}


def process_any_resolution_image_device(image, processor, grid_pinpoints, device="cuda:0") -> torch.Tensor:
    """Same contract as `process_any_resolution_image`, pixel work on the GPU (csrc/preprocess.hip): uint8 RGB image
    (PIL / numpy / uint8 tensor [H, W, 3]) -> bf16 [1 + n_crops, 3, 336, 336] on `device`, bit-identical to the bf16 cast
    of the host function's result (Pillow's 8-bit LANCZOS arithmetic is integer and reproduced exactly)."""
    from vz_hip.preprocess import AnyresPreprocessor
    side = processor.crop_size["height"] if processor is not None else 336
    mean = tuple(getattr(processor, "image_mean", None) or (0.48145466, 0.4578275, 0.40821073))
    std = tuple(getattr(processor, "image_std", None) or (0.26862954, 0.26130258, 0.27577711))
    key = (str(device), side, mean, std)
    if key not in _DEVICE_PRE:
        _DEVICE_PRE[key] = AnyresPreprocessor(device, side, mean, std)
    return _DEVICE_PRE[key](image, grid_pinpoints)


def unpad_image(image_tensor: torch.Tensor, original_size: Tuple[int, int]) -> torch.Tensor:
    """inverse of the letterboxing on a [C, H, W]-like feature map (only reachable from the `unpad` merge types)."""
    ow, oh = original_size
    cw, ch = image_tensor.shape[1:]
    if ow / oh > cw / ch:
        pad = (ch - int(oh * (cw / ow))) // 2
        return image_tensor[:, pad:ch - pad, :]
    pad = (cw - int(ow * (ch / oh))) // 2
    return image_tensor[:, :, pad:cw - pad]
