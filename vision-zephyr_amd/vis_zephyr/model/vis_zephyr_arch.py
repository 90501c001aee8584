"""Multimodal glue of the MI355X Vision-Zephyr build: `encode_images` and
`prepare_inputs_labels_for_multimodal` with the reference's signatures and return contract
(ref:vis_zephyr/model/vis_zephyr_arch.py:107-333,396-530), restructured around the native engine.

The reference walks the batch in Python, calling `embed_tokens` / `torch.cat` per text chunk.  Here the
host only does the *index* arithmetic - it turns (input_ids, attention_mask) into one row map
`(kind, index)` per output position - and a single gather kernel (`vz_embed_splice`) writes the whole
`[B, Smax, 4096]` tensor: coalesced 8 KiB rows, no per-chunk launches.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from ..constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .multimodal_projector.builder import build_multimodal_projector
from .vision_encoder.builder import build_vision_tower


class VisZephyrMetaModel:
    """Owns the vision tower and the projector facades (ref vis_zephyr_arch.py:22-47)."""

    def _init_vision(self, config, owner):
        self.vision_tower = None
        self.mm_projector = None
        if hasattr(config, "mm_vision_tower") and getattr(config, "mm_vision_tower") is not None:
            self.vision_tower = build_vision_tower(config, owner=owner, delay_load=True)
            self.mm_projector = build_multimodal_projector(config, owner=owner)

    def get_vision_tower(self):
        vt = getattr(self, "vision_tower", None)
        return vt[0] if isinstance(vt, list) else vt


def _compact_samples(text_embeddings, tile_sample):
    """a subset of the tiles references a subset of the samples: hand the Q-Former only those rows of the per-sample text table
    (the engine wants n_samples <= tiles) with the map renumbered."""
    used = sorted(set(tile_sample))
    if len(used) == text_embeddings.shape[0]:
        return text_embeddings, tile_sample
    renum = {v: i for i, v in enumerate(used)}
    idx = torch.tensor(used, device=text_embeddings.device)
    return text_embeddings.index_select(0, idx), [renum[v] for v in tile_sample]


class VisZephyrMetaForCausalLM:
    """Mixin with the multimodal entry points; the concrete class provides `engine`, `arch`, `config`,
    `device`, `get_model()`."""

    def get_vision_tower(self):
        return self.get_model().get_vision_tower()

    # a5 ------------------------------------------------------------------------------------------
    def encode_images(self, images, text_embeddings, tile_sample: Optional[Sequence[int]] = None):
        """images [T,3,336,336], text_embeddings [T,Lmax,4096] (or [n_samples,Lmax,4096] together with
        `tile_sample`) -> [T,32,4096]   (ref vis_zephyr_arch.py:120-124)."""
        eng = getattr(self, "engine", None)
        if hasattr(self, "_ensure_ready"):
            self._ensure_ready()
        if eng is not None and eng.tp_size > 1 and getattr(self, "tile_data_parallel", True):
            return self._encode_images_tile_dp(images, text_embeddings, tile_sample)
        T = int(images.shape[0])
        cap = eng.max_tiles if eng is not None else T
        if T <= cap:
            feats = self.get_model().get_vision_tower()(images)
            return self.get_model().mm_projector(feats, text_embeddings=text_embeddings, tile_sample=tile_sample)
        # more tiles than the engine's workspace holds at once (Stage-1 shape: 64 images x up to 5 tiles): tiles are independent
        # units, so they go through in chunks of `max_tiles`; a chunk carries its own rows of the text table / tile -> sample map
        outs = []
        for t0 in range(0, T, cap):
            t1 = min(T, t0 + cap)
            if text_embeddings is None:
                text, ts = None, None
            elif tile_sample is None:
                text, ts = text_embeddings[t0:t1], None
            else:
                text, ts = _compact_samples(text_embeddings, [int(v) for v in tile_sample[t0:t1]])
            feats = self.get_model().get_vision_tower()(images[t0:t1])
            outs.append(self.get_model().mm_projector(feats, text_embeddings=text, tile_sample=ts))
        return torch.cat(outs, dim=0)

    def _encode_images_tile_dp(self, images, text_embeddings, tile_sample):
        """Every tile is an independent unit through CLIP, fusion and the Q-Former (the Q-Former conditions a tile on its
        own sample's text only, ref vis_zephyr_arch.py:163-176): rank r of the tensor-parallel group encodes tiles
        r, r+tp, r+2tp, ... and ONE RCCL all-gather of [ceil(T/tp),32,4096] bf16 per rank hands every rank all visual
        tokens.  Every rank takes part in the collective, also with no tile of its own."""
        from vz_hip import tp as tp_plan
        eng = self.engine
        T = int(images.shape[0])
        mine = tp_plan.local_tiles(T, eng.tp_rank, eng.tp_size)
        per = tp_plan.tiles_per_rank(T, eng.tp_size)
        send = torch.zeros(per, self.arch.qf_queries, self.arch.hidden, dtype=torch.bfloat16, device=self.device)
        if mine:
            idx = torch.tensor(mine, device=images.device)
            if text_embeddings is None:
                text, ts = None, None
            elif tile_sample is None:                    # one text block per tile
                text, ts = text_embeddings.index_select(0, idx.to(text_embeddings.device)), None
            else:                                        # one text block per sample + the tile -> sample map
                text, ts = _compact_samples(text_embeddings, [int(tile_sample[t]) for t in mine])
            feats = self.get_model().get_vision_tower()(images.index_select(0, idx))
            send[: len(mine)] = self.get_model().mm_projector(feats, text_embeddings=text, tile_sample=ts)
        got = eng.all_gather(send).view(eng.tp_size * per, self.arch.qf_queries, self.arch.hidden)
        return got.index_select(0, tp_plan.gathered_index(T, eng.tp_size).to(got.device))

    # a6 / a7 -------------------------------------------------------------------------------------
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, images_size=None):
        vision_tower = self.get_vision_tower()
        if vision_tower is None or images is None or input_ids.shape[1] == 1:        # ref :148-149
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if not (isinstance(images, (list, tuple)) or images.ndim == 5):
            # the reference's 4-D branch feeds 2-D text embeddings to the Q-Former and fails inside torch.cat
            # (ref :209-212)
            raise NotImplementedError("`images` must be a list of [N,3,H,W] tensors or a 5-D tensor [B,N,3,H,W]")
        if hasattr(self, "_ensure_ready"):
            self._ensure_ready()
        merge_type = getattr(self.config, "mm_patch_merge_type", "flat")
        if merge_type != "flat":
            if merge_type.startswith("spatial"):
                raise NotImplementedError("spatial merge types cannot run with the Q-Former projector: the reference "
                                          "itself fails on them (SURVEY.md Appendix A Q5)")
            raise ValueError(f"Unknown mm_patch_merge_type: {merge_type}")

        eng = self.engine
        dev = self.device
        ids_cpu = input_ids.detach().to("cpu", torch.long)
        Bsz, L = ids_cpu.shape
        if isinstance(images, (list, tuple)):
            tiles = [x.unsqueeze(0) if x.ndim == 3 else x for x in images]
        else:
            tiles = [images[i] for i in range(images.shape[0])]
        n_tiles = [int(t.shape[0]) for t in tiles]

        # ---- stage 1: text conditioning per sample (pads included, zero rows up to Lmax: Q3/Q4) ----
        text_ids = [ids_cpu[i][ids_cpu[i] != IMAGE_TOKEN_INDEX] for i in range(len(tiles))]
        Lmax = max(int(t.numel()) for t in text_ids)
        n_s = len(tiles)
        if Lmax > 0:
            kind = torch.full((n_s, Lmax), 2, dtype=torch.int32)
            idx = torch.zeros((n_s, Lmax), dtype=torch.int32)
            for i, t in enumerate(text_ids):
                kind[i, :t.numel()] = 0
                idx[i, :t.numel()] = t.to(torch.int32)
            text = eng.splice(kind.view(-1), idx.view(-1), None).view(n_s, Lmax, -1)
        else:
            text = None
        tile_sample = [s for s, n in enumerate(n_tiles) for _ in range(n)]
        cat_images = torch.cat([t.to(dev) for t in tiles], dim=0)
        feats = self.encode_images(cat_images, text, tile_sample=tile_sample)         # [T,32,H]
        nq = feats.shape[1]
        feat_row0 = [0]
        for n in n_tiles:
            feat_row0.append(feat_row0[-1] + n * nq)                                  # 'flat' merge: [N*32, H] per image

        # ---- stage 2: row map of the spliced sequence ----
        mask_cpu = torch.ones_like(ids_cpu, dtype=torch.bool) if attention_mask is None else \
            attention_mask.detach().to("cpu").bool()
        lab_cpu = None if labels is None else labels.detach().to("cpu", torch.long)
        rows_kind: List[torch.Tensor] = []
        rows_idx: List[torch.Tensor] = []
        rows_lab: List[torch.Tensor] = []
        img_i = 0
        for b in range(Bsz):
            ids = ids_cpu[b][mask_cpu[b]]
            lab = lab_cpu[b][mask_cpu[b]] if lab_cpu is not None else torch.full_like(ids, IGNORE_INDEX)
            is_img = ids == IMAGE_TOKEN_INDEX
            n_img = int(is_img.sum())
            if n_img == 0:                                   # text-only row still consumes a feature slot (Q11)
                rows_kind.append(torch.zeros(ids.numel(), dtype=torch.int32))
                rows_idx.append(ids.to(torch.int32))
                rows_lab.append(lab)
                img_i += 1
                continue
            k_parts, i_parts, l_parts = [], [], []
            cuts = [-1] + torch.where(is_img)[0].tolist() + [ids.numel()]
            for j in range(len(cuts) - 1):
                seg = ids[cuts[j] + 1:cuts[j + 1]]
                k_parts.append(torch.zeros(seg.numel(), dtype=torch.int32))
                i_parts.append(seg.to(torch.int32))
                l_parts.append(lab[cuts[j] + 1:cuts[j + 1]])
                if j < n_img:
                    if img_i >= len(n_tiles):
                        raise IndexError("more image sentinels than images")
                    n = n_tiles[img_i] * nq
                    k_parts.append(torch.ones(n, dtype=torch.int32))
                    i_parts.append(torch.arange(feat_row0[img_i], feat_row0[img_i] + n, dtype=torch.int32))
                    l_parts.append(torch.full((n,), IGNORE_INDEX, dtype=lab.dtype))
                    img_i += 1
            rows_kind.append(torch.cat(k_parts))
            rows_idx.append(torch.cat(i_parts))
            rows_lab.append(torch.cat(l_parts))
        max_len = getattr(self.config, "tokenizer_model_max_length", None)           # ref :308-313
        if max_len is not None:
            rows_kind = [r[:max_len] for r in rows_kind]
            rows_idx = [r[:max_len] for r in rows_idx]
            rows_lab = [r[:max_len] for r in rows_lab]
        # ---- pad + collate (ref :476-530) ----
        Smax = max(int(r.numel()) for r in rows_kind)
        left = getattr(self.config, "tokenizer_padding_side", "right") == "left"
        kind = torch.full((Bsz, Smax), 2, dtype=torch.int32)
        idx = torch.zeros((Bsz, Smax), dtype=torch.int32)
        lab_out = torch.full((Bsz, Smax), IGNORE_INDEX, dtype=torch.long)
        m_out = torch.zeros((Bsz, Smax), dtype=torch.bool)
        p_out = torch.zeros((Bsz, Smax), dtype=torch.long)
        for b in range(Bsz):
            n = int(rows_kind[b].numel())
            if n == 0:
                continue
            sl = slice(Smax - n, Smax) if left else slice(0, n)
            kind[b, sl] = rows_kind[b]
            idx[b, sl] = rows_idx[b]
            lab_out[b, sl] = rows_lab[b]
            m_out[b, sl] = True
            p_out[b, sl] = torch.arange(n)
        embeds = eng.splice(kind.view(-1), idx.view(-1), feats.reshape(-1, feats.shape[-1])).view(Bsz, Smax, -1)

        new_mask = None
        if attention_mask is not None:
            new_mask = m_out.to(attention_mask.dtype).to(attention_mask.device)
        new_pos = None if position_ids is None else p_out.to(position_ids.dtype).to(position_ids.device)
        new_lab = None if labels is None else lab_out.to(labels.dtype).to(labels.device)
        return None, new_pos, new_mask, past_key_values, embeds, new_lab
