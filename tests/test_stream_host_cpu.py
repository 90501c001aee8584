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


class ToyBatchEngine(ToyEngine):
    """+ the static-batch surface of `_generate_batched`: one right-padded prefill, decode_begin with the first tokens"""

    def prefill(self, embeds, seqlens, position_ids=None, all_logits=False, last_logits=True):
        assert not all_logits and last_logits and embeds.shape[0] <= min(self.max_batch, 64)
        self.dec_B = 0
        return None, self.prefill_rows(0, embeds, seqlens)

    def decode_begin(self, first_ids, next_pos, ctx_len):
        self.dec_B = len(next_pos)
        for b in range(self.dec_B):
            assert len(self.rows[b]) == ctx_len[b] == next_pos[b]
        self.cur = [int(t) for t in first_ids.tolist()] + [None] * (self.max_batch - self.dec_B)


@pytest.mark.parametrize("n_seq,max_batch,sync_every,max_new", [(5, 8, 4, 24), (5, 2, 16, 24), (3, 4, 1, 9), (1, 1, 16, 5), (4, 4, 5, 1)])
def test_static_batches_eos_and_padding_equal_solo_generation(monkeypatch, n_seq, max_batch, sync_every, max_new):
    """`_generate_batched`: rows that emitted eos show the pad token from then on, the batch stops at the first step after which
    every row is done, batches beyond the engine's rows run in parts - every row equals the sequence generated alone."""
    from vis_zephyr.model import VisZephyrForCausalLM
    from vz_hip import binding as B
    monkeypatch.setattr(B, "argmax", lambda logits: logits.argmax(-1).to(torch.int32))
    g = torch.Generator().manual_seed(n_seq * 10 + max_batch)
    lens = torch.randint(2, 30, (n_seq,), generator=g).tolist()
    prompts = [torch.randint(1, V, (L,), generator=g).tolist() for L in lens]
    free = [solo(p, max_new, set()) for p in prompts]
    eos = {free[0][min(3, max_new - 1)]} | ({free[-1][0]} if n_seq > 2 else set())     # one ends mid-way, one on its first token
    S = max(lens)
    emb = torch.zeros(n_seq, S, 1, dtype=torch.bfloat16)
    for i, p in enumerate(prompts):
        emb[i, :len(p), 0] = torch.tensor(p, dtype=torch.bfloat16)
    eng = ToyBatchEngine(max_batch)
    me = types.SimpleNamespace(engine=eng, device=torch.device("cpu"))
    me._generate_batched = types.MethodType(VisZephyrForCausalLM._generate_batched, me)
    PAD = 250
    out = me._generate_batched(emb, lens, None, max_new, eos, PAD, sync_every)
    want = [solo(p, max_new, eos) for p in prompts]
    cap = min(max_batch, 64)
    for i0 in range(0, n_seq, cap):              # each part stops when ITS rows are done; the rest of the row is padding
        n_part = max(len(w) for w in want[i0:i0 + cap])
        for i in range(i0, min(n_seq, i0 + cap)):
            row = out[i].tolist()
            assert row[:len(want[i])] == want[i], f"row {i}"
            assert all(t == PAD for t in row[len(want[i]):]) and len(row) >= n_part
    assert out.shape[1] == max(len(w) for w in want)
