"""The one-shot all-reduce of the tensor-parallel decode step (csrc/comm_oneshot.hip: every rank stores its vector into every peer's
receive area as 8-byte {two bf16, sequence tag} granules and sums what arrives in its own area in rank order) - protocol and arithmetic
in ONE process: N receive areas on one GPU stand in for N peers, and the N ranks run as N slices of one launch (co-resident by
construction: N launches on N streams of one GPU may share a hardware queue and then wait for each other until the bounded sweep expires -
seen on the GPU box with 4 streams, and with 2 once the process had other streams open; the production form, one rank per launch and per
GPU, is exercised by the absent-peer test).  No multi-GPU box has been available to the build; over xGMI the
areas are peer-mapped and nothing else changes.  Partition: SURVEY.md section 8e."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def B():
    from vz_hip import binding
    binding.load_library()
    return binding


def _ref(xs):
    """fp32 sum in rank order, rounded to bf16 once (what every rank must produce, bit for bit)"""
    acc = torch.zeros_like(xs[0], dtype=torch.float32)
    for x in xs:
        acc = acc + x.float()
    return acc.to(torch.bfloat16)


@pytest.mark.parametrize("N,n", [(2, 4096), (4, 4096), (8, 4096), (8, 64 * 4096), (3, 770)])
def test_one_shot_all_reduce_over_n_areas(B, N, n):
    dev = "cuda:0"
    cap = 64 * 4096
    areas = [B.oneshot_area(N, cap, dev) for _ in range(N)]
    seqs = [B.oneshot_seq(dev) for _ in range(N)]
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for it in range(5):                                 # consecutive messages: the sequence number advances, the slots alternate
        g = torch.Generator(device="cpu").manual_seed(100 * N + it)
        xs = [(torch.randn(n, generator=g) * (1 + r)).to(torch.bfloat16).to(dev) for r in range(N)]
        outs = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in range(N)]
        B.allreduce_oneshot_all(areas, cap, xs, outs, seqs, err)
        torch.cuda.synchronize()
        assert int(err.item()) == 0
        want = _ref(xs)
        for r in range(N):
            assert torch.equal(outs[r], want), f"N={N} message {it} rank {r}: max |diff| {float((outs[r].float() - want.float()).abs().max())}"
        assert all(int(s[0].item()) == it + 2 and int(s[1].item()) == 0 for s in seqs)
    # in place (the engine reduces the residual stream in place)
    xs = [torch.randn(n).to(torch.bfloat16).to(dev) for _ in range(N)]
    want = _ref(xs)
    B.allreduce_oneshot_all(areas, cap, xs, xs, seqs, err)
    torch.cuda.synchronize()
    assert all(torch.equal(x, want) for x in xs) and int(err.item()) == 0


def test_absent_peer_ends_with_an_error_and_nan(B):
    """a peer that never sends: the sweep is bounded, the launch ends, the error word reads VZ_ASYNC_ONESHOT and the output is NaN (never a
    partial sum)."""
    dev, N, n, cap = "cuda:0", 2, 512, 4096
    areas = [B.oneshot_area(N, cap, dev) for _ in range(N)]
    seq = B.oneshot_seq(dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    x = torch.ones(n, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    B.allreduce_oneshot(areas, 0, cap, x, out, seq, err)          # rank 1 never launches
    torch.cuda.synchronize()
    assert int(err.item()) == B.VZ_ASYNC_ONESHOT
    assert bool(torch.isnan(out.float()).all())
