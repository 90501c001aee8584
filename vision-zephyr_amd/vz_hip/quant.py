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


# ================================================================================================
# NF4 (`load_4bit=True` of the drop-in builder; ref:vis_zephyr/model/builder.py:35-43 -> bitsandbytes, a third-party
# dependency that is not vendored in the reference and not installed here: its published algorithm, QLoRA (Dettmers et al. 2023)
# appendix E / bitsandbytes `quantize_4bit(quant_type="nf4", blocksize=64)`, restated)
#   * the weight is cut into blocks of 64 consecutive elements of a row (in_features is a multiple of 64 for every Zephyr linear, so the
#     flattened blocks bitsandbytes takes never straddle rows, and q / k / v or gate / up quantised apart or stacked give the same codes);
#   * absmax = max|w| of the block; each element w / absmax goes to the NEAREST of the 16 NormalFloat-4 levels below;
#   * dequantised value = level * absmax.
# Not restated: `bnb_4bit_use_double_quant` - bitsandbytes then stores the absmax values themselves in 8 bits (blocks of 256, a dynamic
# exponent code); that perturbs every block's scale by a fraction of a percent.  This build keeps absmax in fp32.  With no bitsandbytes
# in the image there is no vector to pin either part against: parity of this option is "unpinned" (DESIGN.md section 2).
# The engine computes on the DEQUANTISED values rounded to bf16 (`Engine(weight_nf4=True)`): the model is the 4-bit model, the kernels
# are the bf16 ones (no 4-bit weight stream was built; SURVEY section 2.2 names fp8 as this build's quantised form).
# ================================================================================================
NF4_LEVELS = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
              -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
              0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)
NF4_BLOCK = 64


def nf4_quantize(w: torch.Tensor):
    """w [N, K] with K % 64 == 0 -> (codes uint8 [N, K] in 0..15, absmax fp32 [N, K / 64])."""
    N, K = w.shape
    assert K % NF4_BLOCK == 0, "NF4 blocks are 64 consecutive elements of a row"
    x = w.detach().float().reshape(N, K // NF4_BLOCK, NF4_BLOCK)
    absmax = x.abs().amax(dim=2)
    levels = torch.tensor(NF4_LEVELS, dtype=torch.float32, device=w.device)
    mid = (levels[:-1] + levels[1:]) * 0.5                       # nearest level = how many midpoints lie below
    xn = x / torch.clamp(absmax, min=1e-30).unsqueeze(2)
    codes = torch.bucketize(xn, mid, right=False).to(torch.uint8)
    return codes.reshape(N, K), absmax


def nf4_dequantize(codes: torch.Tensor, absmax: torch.Tensor) -> torch.Tensor:
    N, K = codes.shape
    levels = torch.tensor(NF4_LEVELS, dtype=torch.float32, device=codes.device)
    return (levels[codes.long()].reshape(N, K // NF4_BLOCK, NF4_BLOCK) * absmax.unsqueeze(2)).reshape(N, K)


def fake_quantize_nf4(w: torch.Tensor) -> torch.Tensor:
    """w -> level * absmax per 64-element block, fp32."""
    return nf4_dequantize(*nf4_quantize(w))
