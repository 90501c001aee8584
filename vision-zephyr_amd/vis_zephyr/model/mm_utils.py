"""Host-side feeders of the generate path (API of ref:vis_zephyr/model/mm_utils.py:16-186)."""
from __future__ import annotations

from typing import List, Union

import torch
from PIL import Image

from ..constants import IMAGE_TOKEN_INDEX

try:
    from transformers import StoppingCriteria as _StoppingBase
except Exception:                                     # pragma: no cover
    class _StoppingBase:                              # minimal stand-in
        pass


def expand2square(pil_image: Image.Image, background_color) -> Image.Image:
    w, h = pil_image.size
    if w == h:
        return pil_image
    side = max(w, h)
    out = Image.new(pil_image.mode, (side, side), background_color)
    out.paste(pil_image, ((side - w) // 2, (side - h) // 2))
    return out


def process_images(images: Union[Image.Image, List[Image.Image]], image_processor, model_config) -> torch.Tensor:
    single = isinstance(images, Image.Image)
    imgs = [images] if single else list(images)
    mode = getattr(model_config, "aspect_ratio_mode", "square")
    if mode == "pad":
        bg = tuple(int(x * 255) for x in image_processor.image_mean)
        imgs = [expand2square(im, bg) for im in imgs]
    elif mode == "resize":
        side = image_processor.crop_size["height"]
        imgs = [im.resize((side, side), Image.Resampling.LANCZOS) for im in imgs]
    elif mode == "square":
        out = []
        for im in imgs:
            s = min(im.size)
            left, top = int((im.width - s) / 2), int((im.height - s) / 2)
            out.append(im.crop((left, top, left + s, top + s)))
        imgs = out
    px = image_processor(images=imgs, return_tensors="pt")["pixel_values"]
    return px[0] if single else px


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX, return_tensors: str = None):
    """tokenize the text around every `<image>` placeholder and put the sentinel id between the pieces; a BOS emitted by
    the tokenizer is kept once, at the front."""
    chunks = [tokenizer(c).input_ids for c in prompt.split("<image>")]
    ids: List[int] = []
    has_bos = bool(chunks) and bool(chunks[0]) and chunks[0][0] == tokenizer.bos_token_id
    if has_bos:
        ids.append(chunks[0][0])
    skip = 1 if has_bos else 0
    for i, c in enumerate(chunks):
        if i:
            ids.append(image_token_index)
        ids.extend(c[skip:])
    if return_tensors is None:
        return ids
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long)
    raise ValueError(f"Unknown return_tensor type: {return_tensors}")


def get_model_name_from_path(model_path: str) -> str:
    parts = model_path.strip("/").split("/")
    return f"{parts[-2]}_{parts[-1]}" if parts[-1].startswith("checkpoint-") else parts[-1]


class KeywordsStoppingCriteria(_StoppingBase):
    """stop when the generated tail equals one of the keywords' token ids.  As in the reference, ids are compared from
    `input_ids.shape[1]` on (the prompt length at construction), on the CPU, once per generated token."""

    def __init__(self, keywords: List[str], tokenizer, input_ids: torch.LongTensor):
        self.keywords = keywords
        self.keyword_ids = []
        self.max_length = 0
        for kw in keywords:
            ids = tokenizer(kw).input_ids
            if len(ids) > 1 and ids[0] == tokenizer.bos_token_id:
                ids = ids[1:]
            self.max_length = max(self.max_length, len(ids))
            self.keyword_ids.append(torch.tensor(ids))
        self.tokenizer = tokenizer
        self.start_length = input_ids.shape[1]

    def __call__(self, output_ids: torch.LongTensor, scores=None, **kwargs) -> bool:
        for row in output_ids:
            tail = row[self.start_length:].cpu()
            for kid in self.keyword_ids:
                if tail.numel() >= kid.numel() and bool((tail[-kid.numel():] == kid).all()):
                    return True
        return False
