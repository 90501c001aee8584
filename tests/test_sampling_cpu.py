"""oracle/sampling_oracle.py pinned: Philox4x32-10 against the Random123 known-answer vectors, the warper chain against the
installed transformers' own TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper (the reference's third-party
arithmetic on the do_sample path, ref:vis_zephyr/serve/cli.py:171-182), the draw's distribution against softmax."""
import numpy as np
import pytest
import torch

from oracle import sampling_oracle as S


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds: (counter x4, key x2) -> first output word
    assert int(S.philox_word0(0, 0, 0, 0, 0, 0)) == 0x6627E8D5
    assert int(S.philox_word0(*([0xFFFFFFFF] * 6))) == 0x408F276D
    assert int(S.philox_word0(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0)) == 0xD16CFE09


@pytest.mark.parametrize("T,k,p", [(0.2, 50, 1.0), (0.7, 0, 0.9), (1.0, 20, 0.5), (0.1, 50, 0.95), (1.3, 0, 1.0), (0.7, 5, 0.3)])
def test_warp_equals_transformers_warpers(T, k, p):
    tr = pytest.importorskip("transformers")
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(int(T * 100) + k)
    for trial in range(4):
        x = (torch.randn(1, 32000, generator=g) * 2.5).float()
        ref = TemperatureLogitsWarper(T)(None, x.clone())
        if k:
            ref = TopKLogitsWarper(k)(None, ref)
        if p < 1.0:
            ref = TopPLogitsWarper(p)(None, ref)
        mine = S.warp(x[0].numpy(), T, k, p)
        kept_ref = torch.isfinite(ref[0]).numpy()
        kept = np.isfinite(mine)
        # the boundary token of top-p is decided by a cumulative sum (fp32 in HF, fp64 here): allow it to differ only there
        diff = np.flatnonzero(kept_ref != kept)
        assert diff.size <= 1, (T, k, p, diff)
        both = kept & kept_ref
        assert np.array_equal(mine[both], ref[0].numpy()[both])


def test_draw_follows_the_softmax():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(64) * 1.5).astype(np.float32)
    pr = S.probs(x, 0.8, 0, 1.0)
    n = 20000
    cnt = np.zeros(64)
    for c in range(n):
        cnt[S.draw(x, 0.8, 0, 1.0, seed=1234, counter=c)] += 1
    big = pr * n >= 5
    chi2 = float((((cnt - pr * n) ** 2) / (pr * n))[big].sum())
    dof = int(big.sum()) - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)
    assert cnt[~big].sum() <= 5 * max(1.0, (pr[~big] * n).sum()) + 10
