"""VisZephyrForCausalLM for MI355X: the reference's model API
(ref:vis_zephyr/model/language_model/vis_zephyr.py:19-174) on top of the native engine.

The reference class IS an HF `MistralForCausalLM` and delegates generation to `GenerationMixin`.
This one owns a `vz_hip.Engine` (weights + KV cache in HBM, hand-written HIP kernels) and implements
`forward` / `generate` itself with the same observable contract:

  * forward(input_ids|inputs_embeds, images=...) -> object with .loss/.logits/.past_key_values;
    logits for ALL positions, fp32, [B,S,vocab] (what the reference's forward returns).
  * generate(input_ids, images=..., images_size=..., do_sample, temperature, top_p, max_new_tokens,
    eos_token_id, pad_token_id, streamer, stopping_criteria, use_cache) -> LongTensor [B, n_new]
    holding NEW tokens only (HF semantics when generation starts from inputs_embeds; Appendix A Q6).
    streamer.put()/end() and stopping_criteria(ids, scores) are invoked once per token, as HF does.
  * `inputs_embeds=` passed to generate raises NotImplementedError (ref :113-114).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional, Sequence, Union

import torch

from vz_hip.engine import Engine
from vz_hip.synth import ArchConfig

from ..vis_zephyr_arch import VisZephyrMetaForCausalLM, VisZephyrMetaModel

try:                                    # configuration class only - no HF modelling code on this path
    from transformers import MistralConfig as _BaseConfig
except Exception:                       # pragma: no cover - transformers absent
    class _BaseConfig:                  # minimal stand-in with attribute storage
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)


class VisZephyrConfig(_BaseConfig):
    model_type = "vis_zephyr"


def arch_from_config(config) -> ArchConfig:
    g = lambda k, d: getattr(config, k, d) if getattr(config, k, d) is not None else d  # noqa: E731
    hidden = g("hidden_size", 4096)
    heads = g("num_attention_heads", 32)
    return ArchConfig(hidden=hidden, inter=g("intermediate_size", 14336), n_layers=g("num_hidden_layers", 32),
                      n_heads=heads, n_kv_heads=g("num_key_value_heads", 8), head_dim=g("head_dim", hidden // heads),
                      vocab=g("vocab_size", 32000), rms_eps=g("rms_norm_eps", 1e-5),
                      rope_theta=float(g("rope_theta", 10000.0)), sliding_window=g("sliding_window", 4096),
                      # 'cls_patch' keeps the class token: 577 visual tokens per tile (ref vision_encoder.py:66-73)
                      clip_keep_cls=g("mm_vision_select_feature", "patch") == "cls_patch")


class _Embedding:
    """`model.get_model().embed_tokens(ids)` (ref vis_zephyr_arch.py:170,211,248,275)."""

    def __init__(self, owner):
        self._owner = owner

    def __call__(self, ids):
        self._owner._ensure_ready()
        return self._owner.engine.embed_tokens(ids)

    @property
    def weight(self):
        return self._owner.engine.w["llm.embed"]


class _LMHead:
    def __init__(self, owner):
        self._owner = owner

    @property
    def weight(self):
        return self._owner.engine.w["llm.lm_head"]

    def __call__(self, hidden):
        from vz_hip import binding as B
        h = hidden.to(self._owner.device, torch.bfloat16)
        return B.linear(h.reshape(-1, h.shape[-1]).contiguous(), self.weight, out_fp32=True).view(*h.shape[:-1], -1)


class VisZephyrModel(VisZephyrMetaModel):
    """`model.get_model()`: embed_tokens + vision tower + projector (ref vis_zephyr.py:22-26)."""
    config_class = VisZephyrConfig

    def __init__(self, config, owner):
        self.config = config
        self.embed_tokens = _Embedding(owner)
        self._init_vision(config, owner)


class PastKeyValues:
    """The KV cache lives in the engine ([layer][k|v][slot][kv_head][pos][128] bf16); this handle records how
    many positions of each slot are filled."""

    def __init__(self, lengths: Sequence[int], owner=None, epoch: int = 0, continuable: bool = False):
        self.lengths = list(lengths)
        # which model's cache this describes and at which point of its history: the engine owns ONE cache, so a handle is live only
        # until the next forward / generate (forward(past_key_values=...) checks it)
        self.owner, self.epoch, self.continuable = owner, epoch, continuable

    def get_seq_length(self, layer_idx: int = 0):
        return max(self.lengths) if self.lengths else 0


class VisZephyrForCausalLM(VisZephyrMetaForCausalLM):
    config_class = VisZephyrConfig

    def __init__(self, config, device: Union[str, torch.device] = "cuda:0", max_batch: int = 1,
                 max_ctx: int = 4096, max_tiles: int = 8, max_text: int = 2048, engine: Optional[Engine] = None,
                 tp_size: int = 1, tp_rank: int = 0, weight_fp8: bool = False, weight_nf4: bool = False):
        self.config = config
        self.arch = arch_from_config(config)
        self.engine = engine if engine is not None else Engine(self.arch, device=device, max_batch=max_batch,
                                                               max_ctx=max_ctx, max_tiles=max_tiles,
                                                               max_text=max_text, tp_size=tp_size, tp_rank=tp_rank,
                                                               weight_fp8=weight_fp8, weight_nf4=weight_nf4)
        self.device = self.engine.device
        self.dtype = torch.bfloat16
        self.model = VisZephyrModel(config, self)
        self.lm_head = _LMHead(self)
        # hf:generation/configuration_utils.py (pinned 4.52.4) defaults that the reference's callers rely on without naming them:
        # do_sample=True with only `temperature` passed (ref:vis_zephyr/serve/cli.py:171-182) still goes through TopKLogitsWarper(50)
        self.generation_config = SimpleNamespace(eos_token_id=getattr(config, "eos_token_id", 2),
                                                 pad_token_id=getattr(config, "pad_token_id", None), top_k=50, top_p=1.0,
                                                 temperature=1.0)
        self._ring = None
        self._kv_epoch = 0            # bumped by everything that rewrites the engine's KV cache: cache handles of earlier calls go stale

    # ---- construction helpers -------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config, seed: int = 0, **kw):
        m = cls(config, **kw)
        m.engine.load_synthetic(seed)
        m.engine.init_comm()
        if m.get_vision_tower() is not None:
            m.get_vision_tower().is_loaded = True
        return m

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, low_cpu_mem_usage=True, torch_dtype=None,
                        device_map=None, device="cuda:0", **kw):
        """HF's entry point with the reference's usage (ref:vis_zephyr/model/builder.py:108-129): a directory or a hub id
        resolved through the LOCAL cache; `config=` overrides the directory's config.json (the base + mm_projector.bin mode
        passes the finetuned config while the weights come from the Zephyr base).  Streams every weight file of the directory
        into the engine; what the directory lacks (mm_projector.bin arrives through `load_state_dict`, the CLIP tower through
        `get_vision_tower().load_model()`) is completed later - the engine is finalized at first use."""
        import json
        import os
        from vz_hip import weights as W
        path = W.resolve_hub_path(pretrained_model_name_or_path, "pretrained_model_name_or_path")
        if config is None:
            with open(os.path.join(path, "config.json")) as f:
                d = json.load(f)
            for k in ("model_type", "architectures", "transformers_version"):
                d.pop(k, None)
            config = VisZephyrConfig(**d)
        if isinstance(device_map, dict) and "" in device_map:          # ref builder.py:29-31: device_map = {"": device}
            device = device_map[""]
        dev = "cuda:0" if str(device) == "cuda" else device
        engine_kw = {k: kw.pop(k) for k in ("max_batch", "max_ctx", "max_tiles", "max_text", "tp_size", "tp_rank", "weight_fp8", "weight_nf4") if k in kw}
        model = cls(config, device=dev, **engine_kw)
        for name, t in W.iter_backbone(path):
            model.engine.add_weight(name, t)
        return model

    def load_state_dict(self, state_dict, strict: bool = True):
        """`model.load_state_dict(mm_projector_weights, strict=False)` (ref builder.py:121-123): reference-named tensors into
        the engine.  Returns HF's (missing_keys, unexpected_keys) shape with the keys the engine does not take."""
        from vz_hip import weights as W
        unexpected = [k for k, v in W.normalize_keys(state_dict.items()) if not self.engine.add_weight(k, v)]
        if strict and unexpected:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected[:8]}")
        return SimpleNamespace(missing_keys=[], unexpected_keys=unexpected)

    def _ensure_ready(self):
        """finalize lazily: weights may arrive in several steps (from_pretrained, load_state_dict, the tower's load_model);
        the C side names the first missing tensor if something never arrived."""
        if not self.engine.ready:
            self.engine.finalize()
            self.engine.init_comm()

    def load_state_dict_stream(self, named_tensors):
        """feed (reference key, tensor) pairs - e.g. safetensors shards + mm_projector.bin
        (ref:vis_zephyr/model/builder.py:102-138)."""
        self.engine.load_weights(named_tensors)

    # nn.Module-flavoured no-ops the reference's callers invoke
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def half(self):
        return self

    def get_model(self):
        return self.model

    def resize_token_embeddings(self, n: int):
        """ref:vis_zephyr/model/builder.py:141-153 grows embed_tokens / lm_head by `<im_patch>`; new rows are
        the mean of the old ones (HF mean-resizing)."""
        old = self.arch.vocab
        if n is None or n == old:
            return self.model.embed_tokens
        self.engine.resize_vocab(int(n))
        self.arch = self.engine.cfg
        self.config.vocab_size = int(n)
        return self.model.embed_tokens

    # ---- forward (a2) -----------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, images=None, images_size=None, return_dict=None, **kwargs):
        self._ensure_ready()
        if past_key_values is not None:
            return self._forward_continue(input_ids, attention_mask, position_ids, past_key_values, inputs_embeds, labels, images)
        if inputs_embeds is None:
            (input_ids, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                          labels, images, images_size)
        if inputs_embeds is None:
            inputs_embeds = self.get_model().embed_tokens(input_ids)
        Bsz, S = inputs_embeds.shape[0], inputs_embeds.shape[1]
        inputs_embeds, rp_mask, position_ids, shifts = self._to_right_padded(inputs_embeds, attention_mask, position_ids)
        seqlens = self._seqlens(rp_mask, Bsz, S)
        logits = torch.empty(Bsz, S, self.arch.vocab, dtype=torch.float32, device=self.device)
        mb = self.engine.max_batch
        for b0 in range(0, Bsz, mb):        # the engine owns `max_batch` KV slots; larger batches run in chunks
            sl = slice(b0, min(Bsz, b0 + mb))
            la, _ = self.engine.prefill(inputs_embeds[sl], seqlens[sl], None if position_ids is None else position_ids[sl],
                                        all_logits=True, last_logits=False)
            logits[sl] = la
        if shifts is not None:              # back to the caller's (left-padded) column order
            logits = torch.stack([torch.roll(logits[b], shifts[b], 0) for b in range(Bsz)])
        loss = None
        if labels is not None:
            from vz_hip import binding as B
            lab = labels.to(logits.device)
            loss, _ = B.causal_lm_loss(logits.contiguous(), lab)      # hf ForCausalLMLoss on the device (shift, ignore -100, mean over valid)
        self._kv_epoch += 1
        return SimpleNamespace(loss=loss, logits=logits, past_key_values=PastKeyValues(seqlens, self, self._kv_epoch, shifts is None and Bsz <= mb),
                               hidden_states=None, attentions=None)

    def _forward_continue(self, input_ids, attention_mask, position_ids, past_key_values, inputs_embeds, labels, images):
        """forward(input_ids [B, T], past_key_values=<handle of the previous forward>): HF's step API (hf:models/mistral/modeling_mistral.py
        forward with a cache).  The KV cache is the engine's, so only the handle of the LATEST forward on this model is live; the new
        tokens run as decode steps on top of it (teacher-forced), logits [B, T, V] and a handle T positions longer come back."""
        if not isinstance(past_key_values, PastKeyValues) or past_key_values.owner is not self or past_key_values.epoch != self._kv_epoch:
            raise ValueError("forward(past_key_values=...): this cache handle is stale - the engine owns ONE cache, and a later forward / "
                             "generate call has overwritten it")
        if not past_key_values.continuable:
            raise NotImplementedError("forward(past_key_values=...) after a left-padded or chunked batch: re-run the prompt right-padded")
        if images is not None or inputs_embeds is not None or labels is not None or input_ids is None:
            raise NotImplementedError("forward(past_key_values=...) takes new input_ids only (the decode step embeds token ids on the device)")
        ids = input_ids.to(self.device)
        Bsz, T = ids.shape
        if Bsz != len(past_key_values.lengths):
            raise ValueError(f"forward(past_key_values=...): {Bsz} rows of new tokens for a cache of {len(past_key_values.lengths)} rows")
        if attention_mask is not None and not bool(attention_mask[:, -T:].bool().all()):
            raise NotImplementedError("forward(past_key_values=...): padded new tokens")
        lens = list(past_key_values.lengths)
        if max(lens) + T > self.engine.max_ctx:
            raise ValueError(f"forward(past_key_values=...): {max(lens) + T} positions exceed max_ctx = {self.engine.max_ctx}")
        out = torch.empty(Bsz, T, self.arch.vocab, dtype=torch.float32, device=self.device)
        for t in range(T):
            pos = [int(position_ids[b, t]) for b in range(Bsz)] if position_ids is not None else [l + t for l in lens]
            self.engine.decode_begin(ids[:, t].to(torch.int32).contiguous(), pos, [l + t for l in lens])
            _, lg = self.engine.decode_steps(1, return_logits=True)
            out[:, t] = lg[0]
        self._kv_epoch += 1
        return SimpleNamespace(loss=None, logits=out, past_key_values=PastKeyValues([l + T for l in lens], self, self._kv_epoch, True),
                               hidden_states=None, attentions=None)

    __call__ = forward

    @staticmethod
    def _seqlens(attention_mask, Bsz, S) -> List[int]:
        if attention_mask is None:
            return [S] * Bsz
        m = attention_mask.bool().cpu()
        lens = m.sum(1).tolist()
        for b in range(Bsz):
            if not bool(m[b, :lens[b]].all()):
                raise NotImplementedError("attention mask is neither right- nor left-padded (holes are not built)")
        return [max(1, int(x)) for x in lens]

    @staticmethod
    def _to_right_padded(inputs_embeds, attention_mask, position_ids):
        """The engine's KV cache holds every sequence from slot 0 (right padding).  A left-padded batch
        (`tokenizer_padding_side = "left"`, ref:vis_zephyr/model/vis_zephyr_arch.py:515-521) is rolled so that each row's
        real tokens start at column 0; returns the shifts so that per-position outputs can be rolled back."""
        if attention_mask is None:
            return inputs_embeds, attention_mask, position_ids, None
        m = attention_mask.bool()
        Bsz, S = m.shape
        lens = m.sum(1)
        is_left = [bool(lens[b] < S and m[b, S - int(lens[b]):].all() and not m[b, 0]) for b in range(Bsz)]
        if not any(is_left):
            return inputs_embeds, attention_mask, position_ids, None
        shifts = [S - int(lens[b]) if is_left[b] else 0 for b in range(Bsz)]
        emb = torch.stack([torch.roll(inputs_embeds[b], -shifts[b], 0) for b in range(Bsz)])
        msk = torch.stack([torch.roll(attention_mask[b], -shifts[b], 0) for b in range(Bsz)])
        pos = None if position_ids is None else torch.stack([torch.roll(position_ids[b], -shifts[b], 0) for b in range(Bsz)])
        return emb, msk, pos, shifts

    # ---- generate (a3, a13) -------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, images=None, images_size=None, **kwargs):
        self._ensure_ready()
        self._kv_epoch += 1
        position_ids = kwargs.pop("position_ids", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported in this generate function.")
        if images is not None:
            (_, position_ids, attention_mask, _, inputs_embeds, _) = self.prepare_inputs_labels_for_multimodal(
                input_ids, position_ids, attention_mask, None, None, images, images_size)
            if inputs_embeds is None:       # single-token prompt: the reference's early-out returns the ids untouched
                inputs_embeds = self.get_model().embed_tokens(input_ids)
        else:
            inputs_embeds = self.get_model().embed_tokens(input_ids)
        return self._generate_from_embeds(inputs_embeds, attention_mask, position_ids, **kwargs)

    def _generate_from_embeds(self, inputs_embeds, attention_mask, position_ids, max_new_tokens: Optional[int] = None,
                              max_length: Optional[int] = None, do_sample: bool = False, temperature: float = 1.0,
                              top_p: Optional[float] = None, top_k: Optional[int] = -1, eos_token_id=None,
                              pad_token_id=None, streamer=None, stopping_criteria=None, use_cache: bool = True,
                              num_beams: int = 1, generator: Optional[torch.Generator] = None, seed: Optional[int] = None,
                              sync_every: int = 16, timing: Optional[dict] = None, **unused):
        if num_beams != 1:
            raise NotImplementedError("beam search is not on the reference's inference path (cli/eval use sampling/greedy)")
        Bsz, S = inputs_embeds.shape[0], inputs_embeds.shape[1]
        if max_new_tokens is None:
            max_new_tokens = 20 if max_length is None else max(1, max_length - S)
        if eos_token_id is None:
            eos_token_id = self.generation_config.eos_token_id
        eos = set([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id or []))
        if pad_token_id is None:
            pad_token_id = self.generation_config.pad_token_id
        if pad_token_id is None:
            pad_token_id = min(eos) if eos else 0
        greedy = (not do_sample) or temperature is None or temperature <= 0
        if top_k == -1:                      # not passed: HF's generation-config default (50); None / 0 = no top-k filter
            top_k = self.generation_config.top_k
        if top_p is None:
            top_p = self.generation_config.top_p
        if seed is None:                     # one 64-bit seed per generate call: from the caller's generator, else from torch's global one
            seed = 0 if greedy else int(torch.randint(0, 2 ** 62, (1,), generator=generator,
                                                      device=generator.device if generator is not None else "cpu").item())
        inputs_embeds, attention_mask, position_ids, _ = self._to_right_padded(inputs_embeds, attention_mask, position_ids)
        seqlens = self._seqlens(attention_mask, Bsz, S)
        if Bsz > 1 and greedy and streamer is None and stopping_criteria is None:
            return self._generate_batched(inputs_embeds, seqlens, position_ids, max_new_tokens, eos, pad_token_id, sync_every, timing)
        outs = []
        for b in range(Bsz):               # per-token host callbacks / sampling: one sequence at a time
            outs.append(self._generate_one(inputs_embeds[b:b + 1, :seqlens[b]],
                                           None if position_ids is None else position_ids[b:b + 1, :seqlens[b]],
                                           max_new_tokens, greedy, temperature, top_p, top_k, eos, streamer if Bsz == 1 else None,
                                           stopping_criteria, seed + b, sync_every, timing))
        n = max(len(o) for o in outs)
        res = torch.full((Bsz, n), pad_token_id, dtype=torch.long, device=self.device)
        for b, o in enumerate(outs):
            res[b, :len(o)] = torch.tensor(o, dtype=torch.long, device=self.device)
        return res

    def _generate_batched(self, embeds, seqlens, position_ids, max_new, eos, pad_token_id, sync_every, timing=None) -> torch.Tensor:
        """Greedy decoding of up to `max_batch` (<= 64) sequences at once: one right-padded prefill, then every decode step
        streams the weights once for all rows (the KV cache, positions and lengths are per slot).  Rows that hit eos keep
        their slot but emit `pad_token_id` from then on, as HF does."""
        eng = self.engine
        Bsz, S = embeds.shape[0], embeds.shape[1]
        cap = min(eng.max_batch, 64)
        if Bsz > cap:
            parts = [self._generate_batched(embeds[i:i + cap], seqlens[i:i + cap],
                                            None if position_ids is None else position_ids[i:i + cap], max_new, eos,
                                            pad_token_id, sync_every) for i in range(0, Bsz, cap)]
            n = max(p.shape[1] for p in parts)
            out = torch.full((Bsz, n), pad_token_id, dtype=torch.long, device=self.device)
            r = 0
            for p in parts:
                out[r:r + p.shape[0], :p.shape[1]] = p
                r += p.shape[0]
            return out
        if S + max_new > eng.max_ctx:
            raise ValueError(f"prompt ({S}) + max_new_tokens ({max_new}) exceeds the engine's max_ctx ({eng.max_ctx})")
        from vz_hip import binding as B
        _, last = eng.prefill(embeds, seqlens, position_ids, all_logits=False, last_logits=True)
        first = B.argmax(last)
        next_pos = [int(seqlens[b]) if position_ids is None else int(position_ids[b, seqlens[b] - 1]) + 1 for b in range(Bsz)]
        first_cpu = first.to(torch.long).cpu()
        if timing is not None:
            import time
            timing["t_first_token"] = time.perf_counter()        # the host holds every row's first token here
        cols = [first_cpu.view(Bsz, 1)]
        eos_t = torch.tensor(sorted(eos), dtype=torch.long)
        done = torch.isin(first_cpu, eos_t) if eos else torch.zeros(Bsz, dtype=torch.bool)
        eng.decode_begin(first, next_pos, list(seqlens))
        remaining = max_new - 1
        while remaining > 0 and not bool(done.all()):
            n = min(sync_every, remaining) if eos else remaining
            chunk = eng.decode_steps(n).to(torch.long).cpu()             # [B, n]: the only host sync of the chunk
            remaining -= n
            if eos:
                hit = torch.isin(chunk, eos_t)                           # eos emitted at (row, step)
                # a row is finished after its first eos (inclusive): later steps show the pad token
                after = (torch.cumsum(hit.to(torch.int32), dim=1) - hit.to(torch.int32)) > 0
                chunk = torch.where(done.view(Bsz, 1) | after, torch.full_like(chunk, pad_token_id), chunk)
                done_at = done.view(Bsz, 1) | (torch.cumsum(hit.to(torch.int32), dim=1) > 0)      # state after each step
                all_done = done_at.all(dim=0)
                if bool(all_done.any()):                                 # stop at the first step after which every row is done
                    chunk = chunk[:, : int(torch.nonzero(all_done)[0]) + 1]
                done = done_at[:, chunk.shape[1] - 1]
            cols.append(chunk)
        return torch.cat(cols, dim=1).to(self.device)

    # ---- continuous batching: the 23 K-item eval loop without waiting for a batch's longest answer ----
    @torch.no_grad()
    def generate_stream(self, requests, max_new_tokens: int = 128, eos_token_id=None, rows: Optional[int] = None, sync_every: int = 16,
                        admit: Optional[int] = None):
        """Greedy generation over an iterable of requests - dicts with `input_ids` [1, L] (IMAGE_TOKEN_INDEX sentinels allowed),
        optional `images` ([N,3,336,336]) / `images_size`, optional `max_new_tokens` - with CONTINUOUS batching: up to `rows`
        KV-cache rows decode together; a row whose sequence ends (eos or its token budget) is re-armed with the next request at
        the next host sync (every `sync_every` steps), the other rows keep decoding.  Yields `(index, LongTensor[n_new])` in
        completion order.  Rows are independent sequences: every sequence gets the tokens `generate` gives it alone UP TO kernel-route
        re-association - the number of rows decoding together (GEMV / MFMA weight stream / tile GEMM by row count) and, for a shared
        admission prefill, the padded group size (split-K factor of the tile GEMM, fp8 MFMA threshold of a prefill_fp8 engine) choose
        different but equally accurate bf16 evaluation orders, so a near-tie between two candidates can flip.  With the route choice
        pinned (vz_tune_set(26, 0)) the ids are bit-identical (tests/test_stages_gpu.py::test_continuous_batching_matches_static_batches);
        at the default knobs they agree up to the first near-tie (::test_continuous_batching_default_knobs_near_tie).

        Admissions are BATCHED when the engine has cache rows to spare (`max_batch > rows`): all requests entering at one sync
        share one Zephyr prefill (right-padded, up to `admit` = min(max_batch - rows, 16) sequences) into the spare rows, one
        argmax readback, and their KV is then moved to the freed rows (`vz_llm_kv_move_rows`).  The vision stage stays per request:
        the Q-Former's text conditioning makes a request's visual tokens depend on the padded length of its batch."""
        from vz_hip import binding as B
        self._kv_epoch += 1
        self._ensure_ready()
        eng = self.engine
        if eos_token_id is None:
            eos_token_id = self.generation_config.eos_token_id
        eos = set() if eos_token_id is None else ({int(eos_token_id)} if isinstance(eos_token_id, int) else {int(t) for t in eos_token_id})
        n_rows = min(eng.max_batch, 64) if rows is None else int(rows)
        if not 1 <= n_rows <= min(eng.max_batch, 64):
            raise ValueError(f"rows must be in [1, {min(eng.max_batch, 64)}]")
        spare = eng.max_batch - n_rows
        width = min(spare, 16) if admit is None else int(admit)
        if not 0 <= width <= min(spare, 16):
            raise ValueError(f"admit must be in [0, {min(spare, 16)}] (max_batch {eng.max_batch} - rows {n_rows} spare cache rows)")
        it = iter(enumerate(requests))
        slots = [None] * n_rows                 # per row: [request index, tokens so far, budget]
        eng.decode_begin(torch.zeros(n_rows, dtype=torch.int32), [0] * n_rows, [0] * n_rows)       # every row parked
        exhausted = False

        def embed(idx, req):
            ids = req["input_ids"]
            ids = ids if ids.dim() == 2 else ids.unsqueeze(0)
            images = req.get("images")
            budget = int(req.get("max_new_tokens", max_new_tokens))
            if images is not None:
                images = images if isinstance(images, (list, tuple)) else [images]
                emb = self.prepare_inputs_labels_for_multimodal(ids.to(self.device), None, None, None, None, images,
                                                                req.get("images_size"))[4]
            else:
                emb = eng.embed_tokens(ids.to(self.device))
            S = emb.shape[1]
            if S + budget > eng.max_ctx:
                raise ValueError(f"request {idx}: prompt ({S}) + max_new_tokens ({budget}) exceeds the engine's max_ctx ({eng.max_ctx})")
            return emb, S, budget

        while True:
            while not exhausted:                # admit requests into free rows
                free = [r for r in range(n_rows) if slots[r] is None]
                if not free:
                    break
                group = []
                while len(group) < (min(len(free), width) if width >= 2 else 1):
                    try:
                        idx, req = next(it)
                    except StopIteration:
                        exhausted = True
                        break
                    group.append((idx,) + embed(idx, req))
                if not group:
                    break
                if len(group) == 1:             # straight into the free row
                    idx, emb, S, budget = group[0]
                    firsts = [int(B.argmax(eng.prefill_rows(free[0], emb, [S]))[0])]
                else:                           # one prefill for the group in the spare rows, one readback
                    Smax = max(g[2] for g in group)
                    pad = torch.zeros(len(group), Smax, group[0][1].shape[-1], dtype=torch.bfloat16, device=self.device)
                    for j, g in enumerate(group):
                        pad[j, :g[2]] = g[1][0]
                    firsts = B.argmax(eng.prefill_rows(n_rows, pad, [g[2] for g in group])).tolist()
                moves = []
                for j, (idx, emb, S, budget) in enumerate(group):
                    first = int(firsts[j])
                    if first in eos or budget <= 1:
                        yield idx, torch.tensor([first], dtype=torch.long)
                        continue
                    r = free.pop(0)
                    if len(group) > 1:
                        moves.append((n_rows + j, r, S))
                    slots[r] = [idx, [first], budget]
                    group[j] = (idx, None, S, budget, r, first)
                if moves:
                    eng.kv_move_rows([m[0] for m in moves], [m[1] for m in moves], [m[2] for m in moves])
                for g in group:
                    if len(g) == 6:
                        eng.decode_set_row(g[4], g[5], g[2], g[2])
            if all(s is None for s in slots):
                return
            n = min([sync_every] + [s[2] - len(s[1]) for s in slots if s is not None])
            chunk = eng.decode_steps(n).cpu()                           # [rows, n]; the only host sync of the chunk
            for r, st in enumerate(slots):
                if st is None:
                    continue
                for t in chunk[r].tolist():
                    st[1].append(int(t))
                    if int(t) in eos or len(st[1]) >= st[2]:
                        break
                if st[1][-1] in eos or len(st[1]) >= st[2]:
                    yield st[0], torch.tensor(st[1], dtype=torch.long)
                    eng.decode_set_row(r, 0, 0, 0)                       # park until the next request arrives
                    slots[r] = None

    def _token_ring(self, n: int = 8) -> torch.Tensor:
        """pinned host buffer the step tails write their tokens to (device-visible: hipHostMalloc memory is mapped)."""
        if self._ring is None or self._ring.numel() != n:
            self._ring = torch.zeros(n, dtype=torch.int32).pin_memory()
        return self._ring

    def _generate_one(self, embeds, position_ids, max_new, greedy, temperature, top_p, top_k, eos, streamer,
                      stopping_criteria, seed, sync_every, timing=None) -> List[int]:
        """One sequence.  The first token comes from the prefill logits (argmax / one draw of the device sampler with counter 0),
        every later one from a replay of the per-token hipGraph whose tail is the argmax or the sampling kernel - also on the path
        `script/run_cli.sh` takes (streamer + stopping criteria + do_sample, ref:vis_zephyr/serve/cli.py:155-182): there the host
        keeps ONE step in flight and reads token t from a host-visible ring when the event behind step t fires, so the callbacks
        of token t run under step t+1 and no logits ever travel."""
        from vz_hip import binding as B
        eng = self.engine
        S = embeds.shape[1]
        if S + max_new > eng.max_ctx:
            raise ValueError(f"prompt ({S}) + max_new_tokens ({max_new}) exceeds the engine's max_ctx ({eng.max_ctx})")
        _, last = eng.prefill(embeds, [S], position_ids, all_logits=False, last_logits=True)
        next_pos = S if position_ids is None else int(position_ids[0, -1]) + 1
        per_token = streamer is not None or stopping_criteria is not None
        out: List[int] = []
        if streamer is not None:
            # HF hands the (empty, since generation starts from embeddings) prompt ids to the streamer first;
            # TextStreamer(skip_prompt=True) swallows exactly one put() as "the prompt"
            streamer.put(torch.empty((1, 0), dtype=torch.long))
        first = B.argmax(last) if greedy else B.sample(last, temperature, top_k, top_p, seed, 0)
        tok = int(first[0])          # int(): the host holds the first token here
        if timing is not None:
            import time
            timing["t_first_token"] = time.perf_counter()
        out.append(tok)
        if self._emit(out, streamer, stopping_criteria, None, eos) or max_new <= 1:
            return self._finish(out, streamer)
        eng.set_sampling(not greedy, temperature if not greedy else 1.0, top_k, top_p, seed, first_counter=1)
        try:
            if per_token:
                ring = self._token_ring()
                R = ring.numel()
                eng.set_ring(ring)
                eng.decode_begin(first, [next_pos], [S])
                dev_out = torch.empty(1, 1, dtype=torch.int32, device=self.device)
                events = {}

                def launch(t):               # step producing generated token index t (draw counter t)
                    eng.decode_steps(1, out=dev_out)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    events[t] = ev

                launch(1)
                t = 1
                while True:
                    if t + 1 < max_new:
                        launch(t + 1)        # speculative: its cache slot / ring slot are simply ignored if token t stops the sequence
                    events.pop(t).synchronize()
                    out.append(int(ring[t % R]))
                    if self._emit(out, streamer, stopping_criteria, None, eos) or len(out) >= max_new:
                        break
                    t += 1
                torch.cuda.current_stream(self.device).synchronize()      # a speculative step may still be running
                eng.check_async()
                return self._finish(out, streamer)
            # no host callbacks: steps are enqueued back to back, the host looks at the ids every `sync_every` tokens only to honour eos
            eng.decode_begin(first, [next_pos], [S])
            remaining = max_new - 1
            while remaining > 0:
                n = min(sync_every, remaining) if eos else remaining
                ids = eng.decode_steps(n)[0].tolist()
                eng.check_async()                 # a bounded device-side wait that expired = invalid ids: fail loudly
                remaining -= n
                for t in ids:
                    out.append(int(t))
                    if int(t) in eos:
                        return out
            return out
        finally:
            eng.set_sampling(False)
            eng.set_ring(None)

    def _emit(self, out, streamer, stopping_criteria, scores, eos) -> bool:
        """per-token host callbacks; returns True when generation must stop."""
        if streamer is not None:
            streamer.put(torch.tensor([out[-1]], dtype=torch.long))
        stop = out[-1] in eos
        if stopping_criteria is not None and not stop:
            ids = torch.tensor([out], dtype=torch.long, device=self.device)
            crits = stopping_criteria if isinstance(stopping_criteria, (list, tuple)) else [stopping_criteria]
            for c in crits:
                r = c(ids, scores)
                if bool(r.all() if torch.is_tensor(r) else r):
                    stop = True
        return stop

    @staticmethod
    def _finish(out, streamer):
        if streamer is not None:
            streamer.end()
        return out

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, inputs_embeds=None, **kwargs):
        """ref vis_zephyr.py:144-170: pass-through that re-attaches images / images_size."""
        images = kwargs.pop("images", None)
        images_size = kwargs.pop("images_size", None)
        inputs = dict(input_ids=input_ids, past_key_values=past_key_values, inputs_embeds=inputs_embeds, **kwargs)
        if images is not None:
            inputs["images"] = images
        if images_size is not None:
            inputs["images_size"] = images_size
        return inputs


# Register the model / config with HF's Auto classes (ref:vis_zephyr/model/language_model/vis_zephyr.py:173-174): lets
# `AutoConfig.from_pretrained` read a `"model_type": "vis_zephyr"` config.json and `AutoModelForCausalLM.from_pretrained(...)`
# dispatch to VisZephyrForCausalLM.from_pretrained above.
try:
    from transformers import AutoConfig, AutoModelForCausalLM
    AutoConfig.register("vis_zephyr", VisZephyrConfig, exist_ok=True)
    AutoModelForCausalLM.register(VisZephyrConfig, VisZephyrForCausalLM, exist_ok=True)
except Exception:                       # pragma: no cover - transformers absent or too old for exist_ok
    pass
