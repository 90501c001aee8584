"""W8A16 weight quantisation for the decode weight stream ("fp8 weights").

Each output row of a linear gets ONE power-of-two scale 2^e (an MX-style e8m0 scale per row) and its weights are rounded
to OCP e4m3 (`torch.float8_e4m3fn`, round-to-nearest-even, |x| <= 448):

    e   = ceil(log2(max|w_row| / 448))          w8 = e4m3(w_row * 2^-e)          w_q = 2^e * w8

Because the scale is a power of two, w_q is EXACTLY representable in bf16 (4 significant bits, exponent inside bf16's range):
the decode GEMV streams the 1-byte rows (half the HBM bytes of bf16), the prefill MFMA GEMMs run on the bf16 tensor holding
w_q, and both compute with one and the same model.  The oracle restates the same rounding on the CPU (`oracle.vz_oracle
.quantize_state_dict`), so parity is against a reference running the identical quantised weights."""
from __future__ import annotations

from typing import Tuple

import torch

E4M3_MAX = 448.0


def row_exponents(w: torch.Tensor) -> torch.Tensor:
    """int32 [N]: e = ceil(log2(amax / 448)) per row (0 for an all-zero row)."""
    amax = w.detach().float().abs().amax(dim=1)
    e = torch.ceil(torch.log2(torch.clamp(amax, min=1e-30) / E4M3_MAX))
    e = torch.where(amax > 0, e, torch.zeros_like(e))
    return e.clamp_(-100, 100).to(torch.int32)


def quantize_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N,K] (any float dtype) -> (w8 uint8 [N,K] holding e4m3 bytes, scale fp32 [N] = 2^e)."""
    e = row_exponents(w)
    scale = torch.exp2(e.float())
    q = (w.detach().float() * torch.exp2(-e.float()).unsqueeze(1)).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def dequantize_rows(w8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """fp32 [N,K] = 2^e * e4m3 (exact; also exactly representable in bf16)."""
    return w8.view(torch.float8_e4m3fn).float() * scale.float().unsqueeze(1)


def fake_quantize_rows(w: torch.Tensor) -> torch.Tensor:
    """w -> w_q in fp32 (what both the engine and the oracle compute with)."""
    return dequantize_rows(*quantize_rows(w))
