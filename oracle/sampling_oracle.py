"""CPU restatement of the sampled decode step's tail.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

What the reference computes (third-party arithmetic, transformers pinned 4.52.4 - `ref:environment.yaml:108`):
`GenerationMixin._sample` with `do_sample=True` (hf:generation/utils.py) takes `logits[:, -1].float()`, applies the
warpers in this order - `TemperatureLogitsWarper` (scores / temperature), `TopKLogitsWarper` (scores < topk(k).values[-1]
-> -inf; `GenerationConfig.top_k` defaults to 50 and the reference's callers never override it:
`ref:vis_zephyr/serve/cli.py:171-182`, `ref:vis_zephyr/eval/eval_vqa.py:190-200`), `TopPLogitsWarper` (sort ascending,
cumulative softmax, drop where cumsum <= 1 - top_p, keep the last) - then `softmax` and `torch.multinomial(probs, 1)`.

PINNING: `warp()` below is checked against the installed transformers' own warper classes on seeded logits
(tests/test_sampling_cpu.py, runs wherever transformers imports).  torch.multinomial's bit stream belongs to torch's
generator, not to the reference, so the draw itself is defined HERE (and in vision-zephyr_amd/csrc/sampling.hip, which must
match it draw for draw): a Gumbel race, token = argmax_i(x_i + G_i) over the kept tokens, G_i = -log(-log(u_i)),
u_i = ((w_i >> 9) + 0.5) * 2^-23, w_i = word 0 of Philox4x32-10(key = seed lo/hi, counter = (i, row, draw counter, 0x565a)).
P[token = i] = softmax(x)_i - the distribution `torch.multinomial` draws from (tests check both, the exact draw and a
chi-square of the device's draws against `probs()`).
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def philox_word0(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10, first output word; array arguments broadcast (uint32 semantics carried in uint64)."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & _M32 for v in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    A, Bm = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = A * c0, Bm * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _M32
        n1 = p1 & _M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _M32
        n3 = p0 & _M32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return c0.astype(np.uint32)


def warp(logits: np.ndarray, temperature: float, top_k: int = 0, top_p: float = 1.0) -> np.ndarray:
    """fp32 logits [V] -> tempered scores with the filtered entries at -inf (HF's warper chain)."""
    x = (logits.astype(np.float32) / np.float32(temperature)).astype(np.float32)
    V = x.shape[0]
    if top_k and 0 < top_k < V:
        kth = np.sort(x)[V - top_k]
        x = np.where(x < kth, -np.inf, x).astype(np.float32)
    if top_p is not None and top_p < 1.0:
        order = np.argsort(x, kind="stable")
        sx = x[order].astype(np.float64)
        e = np.exp(sx - sx[-1])
        cp = np.cumsum(e / e.sum())
        remove = cp <= (1.0 - top_p)
        remove[-1] = False
        # ties with the first kept value stay together (the sort order inside a tie is not defined by the reference)
        if remove.any():
            first_kept = sx[np.argmin(remove)]
            remove &= sx < first_kept
        out = x.copy()
        out[order[remove]] = -np.inf
        x = out
    return x


def probs(logits, temperature, top_k=0, top_p=1.0) -> np.ndarray:
    x = warp(logits, temperature, top_k, top_p).astype(np.float64)
    e = np.exp(x - x.max())
    return e / e.sum()


def gumbel(V: int, row: int, counter: int, seed: int) -> np.ndarray:
    w = philox_word0(np.arange(V), row, counter, 0x565A, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = ((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    return (-np.log(-np.log(u, dtype=np.float32), dtype=np.float32)).astype(np.float32)


def draw(logits: np.ndarray, temperature: float, top_k: int, top_p: float, seed: int, counter: int, row: int = 0,
         return_margin: bool = False):
    """the token the device draws for (seed, draw counter, row) and - for the tests' tie handling - the gap between the two
    best race scores."""
    x = warp(logits, temperature, top_k, top_p)
    s = (x + gumbel(x.shape[0], row, counter, seed)).astype(np.float32)
    s = np.where(np.isneginf(x), -np.inf, s)
    i = int(np.argmax(s))
    if not return_margin:
        return i
    two = np.partition(s, -2)[-2:]
    return i, float(two[1] - two[0])
