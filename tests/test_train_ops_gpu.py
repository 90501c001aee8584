"""Helpers of the Stage-1 training step at op level (train.hip): the 64 x 64 transposition through the hardware transposing LDS read (the
dY^T / X^T / W^T copies of the weight- and input-gradient GEMMs) and the column sums (bias gradients).  Exact: a transposition moves
bits; the column sums are compared with a float64 sum (fp32 accumulation in a fixed order: 1e-6 relative)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from vz_hip import binding
    binding.load_library()
    return binding


@pytest.mark.parametrize("R,C", [(64, 64), (2560, 4096), (200, 136), (8, 8), (10240, 1024), (33, 47), (64, 72)])
def test_transpose_is_exact(B, R, C):
    x = torch.randn(R, C, generator=torch.Generator().manual_seed(R + C)).to("cuda", torch.bfloat16)
    y = B.transpose(x)
    assert torch.equal(y, x.t().contiguous())
    # a strided source (leading dimension wider than C)
    wide = torch.randn(R, C + 16, generator=torch.Generator().manual_seed(1)).to("cuda", torch.bfloat16)
    assert torch.equal(B.transpose(wide[:, :C]), wide[:, :C].t().contiguous())


@pytest.mark.parametrize("rows,cols", [(37, 64), (2560, 4096), (10240, 12288), (5000, 100), (4096, 1024)])
def test_colsum_adds_into_the_output(B, rows, cols):
    y = torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows)).to("cuda", torch.bfloat16)
    out = torch.full((cols,), 0.5, dtype=torch.float32, device="cuda")
    B.colsum(y, out)
    ref = y.double().sum(0) + 0.5
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err <= 1e-6, err
    out2 = torch.full((cols,), 0.5, dtype=torch.float32, device="cuda")
    B.colsum(y, out2)
    assert torch.equal(out, out2)          # fixed order: reproducible
