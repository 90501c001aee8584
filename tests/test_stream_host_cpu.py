"""Host logic of `VisZephyrForCausalLM.generate_stream` (continuous batching with single and batched admissions) on the CPU: the
engine is replaced by a toy whose next token depends on the WHOLE history held in a KV-cache row, so a request admitted into the
wrong row, moved with the wrong length, armed with the wrong position or decoded past its budget changes its tokens.  (The same
loop against the real engine: tests/test_stages_gpu.py::test_continuous_batching_matches_static_batches.)"""
import types

import pytest
import torch

V = 251           # toy vocabulary: every id is exact in bf16


def next_token(history):
    return (sum((i + 1) * t for i, t in enumerate(history)) * 31 + 7 * len(history) + 3) % V


def solo(prompt, budget, eos):
    hist, out = list(prompt), []
    while len(out) < budget:
        t = next_token(hist)
        out.append(t)
        hist.append(t)
        if t in eos:
            break
    return out


class ToyEngine:
    """the engine surface generate_stream touches; cache row r holds the token history of the sequence living there"""

    def __init__(self, max_batch, max_ctx=512):
        self.max_batch, self.max_ctx = max_batch, max_ctx
        self.rows = [None] * max_batch          # history per cache row
        self.cur = [None] * max_batch           # armed input token per decode row (None = parked)
        self.dec_B = 0
        self.prefill_shapes, self.moves = [], []

    def embed_tokens(self, ids):
        return ids.to(torch.bfloat16).unsqueeze(-1)

    def prefill_rows(self, row0, emb, seqlens, position_ids=None):
        k = emb.shape[0]
        assert row0 + k <= self.max_batch
        self.prefill_shapes.append((row0, k, emb.shape[1]))
        logits = torch.full((k, V), -1.0)
        for j in range(k):
            assert self.cur[row0 + j] is None if row0 + j < self.dec_B else True, "prefill into a row that is decoding"
            hist = [int(v) for v in emb[j, :seqlens[j], 0].float().tolist()]
            self.rows[row0 + j] = hist
            logits[j, next_token(hist)] = 1.0
        return logits

    def kv_move_rows(self, src, dst, lens):
        assert len(set(dst)) == len(dst) and not set(dst) & set(src)
        for s_, d_, n in zip(src, dst, lens):
            assert s_ >= self.dec_B and d_ < self.dec_B and self.cur[d_] is None
            self.rows[d_] = list(self.rows[s_][:n])
        self.moves.append((tuple(src), tuple(dst), tuple(lens)))

    def decode_begin(self, first_ids, next_pos, ctx_len):
        self.dec_B = len(next_pos)
        assert all(c == 0 for c in ctx_len)
        self.cur = [None] * self.max_batch

    def decode_set_row(self, row, token, next_pos, ctx_len):
        assert row < self.dec_B
        if ctx_len == 0:
            self.cur[row] = None
            return
        assert len(self.rows[row]) == ctx_len == next_pos, "armed with a context that is not the prefilled one"
        self.cur[row] = token

    def decode_steps(self, n):
        out = torch.zeros(self.dec_B, n, dtype=torch.int32)
        for s_ in range(n):
            for r in range(self.dec_B):
                if self.cur[r] is None:
                    continue
                self.rows[r].append(self.cur[r])
                assert len(self.rows[r]) < self.max_ctx
                self.cur[r] = next_token(self.rows[r])
                out[r, s_] = self.cur[r]
        return out


@pytest.fixture()
def stream(monkeypatch):
    from vis_zephyr.model import VisZephyrForCausalLM
    from vz_hip import binding as B
    monkeypatch.setattr(B, "argmax", lambda logits: logits.argmax(-1).to(torch.int32))

    def run(eng, reqs, **kw):
        me = types.SimpleNamespace(_kv_epoch=0, _ensure_ready=lambda: None, engine=eng, device=torch.device("cpu"),
                                   generation_config=types.SimpleNamespace(eos_token_id=None),
                                   prepare_inputs_labels_for_multimodal=None)
        return dict(VisZephyrForCausalLM.generate_stream.__wrapped__(me, reqs, **kw))
    return run


def _requests(n, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(3, 40, (n,), generator=g).tolist()
    budgets = torch.randint(1, 30, (n,), generator=g).tolist()
    budgets[2] = 1                                                  # finishes on its first token: never takes a row
    prompts = [torch.randint(1, V, (L,), generator=g) for L in lens]
    return [{"input_ids": p.unsqueeze(0), "max_new_tokens": b} for p, b in zip(prompts, budgets)], prompts, budgets


@pytest.mark.parametrize("rows,max_batch,admit,sync_every", [(3, 3, None, 4), (3, 6, None, 4), (3, 6, 2, 16), (3, 6, 0, 1), (1, 4, None, 8),
                                                              (8, 24, None, 4), (5, 7, None, 3)])
def test_stream_equals_solo_generation(stream, rows, max_batch, admit, sync_every):
    reqs, prompts, budgets = _requests(23, seed=rows * 100 + max_batch)
    # an eos id that some sequences emit mid-way and (so that both endings are exercised) that ends request 5 at its first token
    eos = {next_token(prompts[5].tolist()), solo(prompts[7].tolist(), budgets[7] + 3, set())[min(2, budgets[7] - 1)]}
    eng = ToyEngine(max_batch)
    got = stream(eng, reqs, eos_token_id=sorted(eos), rows=rows, sync_every=sync_every, admit=admit)
    assert sorted(got) == list(range(len(reqs)))
    for i, (p, b) in enumerate(zip(prompts, budgets)):
        assert got[i].tolist() == solo(p.tolist(), b, eos), f"request {i}"
    spare = max_batch - rows
    width = min(spare, 16) if admit is None else admit
    if width >= 2 and rows >= 2:
        assert eng.moves, "no batched admission happened"
        assert all(r0 == rows and k <= width for r0, k, _ in eng.prefill_shapes if k > 1)     # groups live in the spare rows only
    elif width < 2:
        assert not eng.moves and all(k == 1 and r0 < rows for r0, k, _ in eng.prefill_shapes)


def test_stream_argument_checks(stream):
    reqs, _, _ = _requests(4, seed=1)
    with pytest.raises(ValueError):
        stream(ToyEngine(4), reqs, rows=5)
    with pytest.raises(ValueError):
        stream(ToyEngine(6), reqs, rows=4, admit=3)                 # two spare rows only
    long = [{"input_ids": torch.ones(1, 500, dtype=torch.long), "max_new_tokens": 20}]
    with pytest.raises(ValueError):
        stream(ToyEngine(2), long, rows=1)                          # prompt + budget beyond max_ctx
