"""Stage-1 pretrain step on the native engine.

What the reference runs per optimiser step (`ref:vis_zephyr/train/train.py:817-829` freezes everything but
`model.get_model().mm_projector`; `ref:vis_zephyr/model/language_model/vis_zephyr.py:51-98` -> HF `ForCausalLMLoss`;
`ref:vis_zephyr/train/vis_zephyr_trainer.py:224-302` builds AdamW with every `mm_projector` parameter in the groups whose lr is
`--mm_projector_lr` = 2e-3 (`ref:script/pretrain.sh:16`; the 2e-5 of `--learning_rate`, `:38`, only reaches non-projector groups,
which are empty in Stage 1), weight decay 0, HF's cosine schedule with 3 % warm-up scaling that 2e-3 (`:39-41`); DeepSpeed ZeRO-2 =
data parallelism with AVERAGED gradients; `_save_checkpoint` (`ref:...vis_zephyr_trainer.py:304-348`) writes `mm_projector.bin`):

    loss = model(input_ids, attention_mask, labels=labels, images=images).loss ; loss.backward() ; optimizer.step()

Here: `Stage1Trainer.step(...)` - the host does the same index logic as `prepare_inputs_labels_for_multimodal` (row map of the
spliced sequence, shifted labels, count of valid targets), the device does everything else in `vz_train_stage1_accumulate`
(forward with saved activations, backward through the frozen Zephyr into the Q-Former, 165 parameter gradients) and
`vz_train_adamw_step`.  Batches larger than the engine's tile / row capacity run as micro-batches that accumulate into the same
fp32 gradient arena (the loss normaliser is the valid-target count of the WHOLE batch, so the sum is exact).
No CPU fallback: this module needs libviszephyr_hip.so and a GPU."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import binding as B

IGNORE_INDEX, IMAGE_TOKEN_INDEX = -100, -200


MM_PROJECTOR_LR = 2e-3       # ref:script/pretrain.sh:16 (--mm_projector_lr): the base rate of the only trainable group of Stage 1


def lr_at(step: int, total_steps: int, base_lr: float = MM_PROJECTOR_LR, warmup_ratio: float = 0.03) -> float:
    """HF `get_cosine_schedule_with_warmup` as the Trainer builds it (ref:script/pretrain.sh:39-41) around the projector groups'
    base rate: linear warm-up over ceil(ratio * total) steps, then half a cosine to zero; `step` = optimiser steps already taken."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


# reference parameter name (under model.mm_projector.) <-> engine tensor (+ row slice of it)
def _ref_to_engine(cfg) -> Dict[str, tuple]:
    H = cfg.hidden
    out = {"learned_queries": ("qf.queries", None), "pre_norm.weight": ("qf.pre_norm.w", None), "pre_norm.bias": ("qf.pre_norm.b", None),
           "norm.weight": ("qf.norm.w", None), "norm.bias": ("qf.norm.b", None)}
    for i in range(cfg.qf_blocks):
        p, q = f"blocks.{i}.", f"qf.{i}."
        for k in (1, 2, 3):
            out[p + f"norm{k}.weight"] = (q + f"n{k}.w", None)
            out[p + f"norm{k}.bias"] = (q + f"n{k}.b", None)
        out[p + "self_attn.in_proj_weight"] = (q + "sa_in.w", None)
        out[p + "self_attn.in_proj_bias"] = (q + "sa_in.b", None)
        out[p + "self_attn.out_proj.weight"] = (q + "sa_out.w", None)
        out[p + "self_attn.out_proj.bias"] = (q + "sa_out.b", None)
        out[p + "cross_attn.q_proj_weight"] = (q + "ca_q.w", None)
        out[p + "cross_attn.k_proj_weight"] = (q + "ca_kv.w", slice(0, H))
        out[p + "cross_attn.v_proj_weight"] = (q + "ca_kv.w", slice(H, 2 * H))
        out[p + "cross_attn.in_proj_bias"] = ((q + "ca_q.b", q + "ca_kv.b"), None)
        out[p + "cross_attn.out_proj.weight"] = (q + "ca_out.w", None)
        out[p + "cross_attn.out_proj.bias"] = (q + "ca_out.b", None)
        out[p + "ffn.0.weight"] = (q + "ffn1.w", None)
        out[p + "ffn.0.bias"] = (q + "ffn1.b", None)
        out[p + "ffn.2.weight"] = (q + "ffn2.w", None)
        out[p + "ffn.2.bias"] = (q + "ffn2.b", None)
    return out


class Stage1Trainer:
    def __init__(self, model):
        self.model = model
        self.eng = model.engine
        model._ensure_ready()
        self.lib = B.load_library()
        h = C.c_void_p()
        B.check(self.lib.vz_train_create(self.eng.h, C.byref(h), self.eng._s()))
        self.h = h
        self.table = {}
        name, n, off, mat = C.c_char_p(), C.c_long(), C.c_long(), C.c_int()
        for i in range(self.lib.vz_train_param_count(self.h)):
            B.check(self.lib.vz_train_param_info(self.h, i, C.byref(name), C.byref(n), C.byref(off), C.byref(mat)))
            self.table[name.value.decode()] = (off.value, n.value, bool(mat.value))
        g, ma, m, v, tot = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_long()
        B.check(self.lib.vz_train_arenas(self.h, C.byref(g), C.byref(ma), C.byref(m), C.byref(v), C.byref(tot)))
        self.total = tot.value
        self._grad_ptr, self._master_ptr = g.value, ma.value
        self.steps_done = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.vz_train_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- views of the flat fp32 arenas (device memory owned by the trainer) ----
    def _view(self, base_ptr: int, name: str) -> torch.Tensor:
        off, n, _ = self.table[name]
        shape = tuple(self.eng.w[name].shape)
        return _from_ptr(base_ptr + 4 * off, n, self.eng.device).view(shape)

    def grad(self, name: str) -> torch.Tensor:
        """fp32 gradient of an ENGINE tensor (e.g. 'qf.3.ffn1.w'), a view of the arena"""
        return self._view(self._grad_ptr, name)

    def master(self, name: str) -> torch.Tensor:
        return self._view(self._master_ptr, name)

    def reference_grads(self) -> Dict[str, torch.Tensor]:
        """gradients under the reference's parameter names (`model.mm_projector.*`), as `named_parameters()` of the reference's
        Q-Former orders its tensors: k_proj / v_proj are the two halves of the engine's stacked cross-attention matrix, the
        cross-attention `in_proj_bias` is q | k | v."""
        out = {}
        for ref, (eng_name, rows) in _ref_to_engine(self.eng.cfg).items():
            if isinstance(eng_name, tuple):
                out["model.mm_projector." + ref] = torch.cat([self.grad(n).reshape(-1) for n in eng_name])
            else:
                g = self.grad(eng_name)
                out["model.mm_projector." + ref] = g if rows is None else g[rows]
        return out

    def set_masters_from_reference(self, named_fp32):
        """exact fp32 starting point of the optimiser (a checkpoint's fp32 projector / the oracle's weights): the engine's bf16
        working copies keep their rounded values."""
        table = _ref_to_engine(self.eng.cfg)
        H = self.eng.cfg.hidden
        for k, v in named_fp32:
            k = k[len("model.mm_projector."):] if k.startswith("model.mm_projector.") else k
            if k not in table:
                continue
            eng_name, rows = table[k]
            v = v.to(self.eng.device, torch.float32)
            if isinstance(eng_name, tuple):
                self.master(eng_name[0]).copy_(v.reshape(-1)[:H])
                self.master(eng_name[1]).copy_(v.reshape(-1)[H:])
            elif rows is None:
                self.master(eng_name).copy_(v.reshape(self.master(eng_name).shape))
            else:
                self.master(eng_name)[rows].copy_(v)

    def zero_grad(self):
        B.check(self.lib.vz_train_zero_grad(self.h, self.eng._s()))

    # ---- one micro-batch: forward + backward, gradients accumulate ----
    def _accumulate(self, ids_cpu, mask_cpu, lab_cpu, tiles: List[torch.Tensor], inv_n: float, Lmax_batch: Optional[int] = None) -> float:
        eng, cfg, dev = self.eng, self.eng.cfg, self.eng.device
        Bsz = ids_cpu.shape[0]
        nq = cfg.qf_queries
        n_tiles = [int(t.shape[0]) for t in tiles]
        text_ids = [ids_cpu[i][ids_cpu[i] != IMAGE_TOKEN_INDEX] for i in range(len(tiles))]      # pads included (Appendix A Q4)
        # the Q-Former's text conditioning is zero-padded to the longest text of the BATCH the reference's forward sees (Appendix A Q3):
        # a micro-batch pads to the whole batch's length, or it would compute a different function
        Lmax = max(int(t.numel()) for t in text_ids) if Lmax_batch is None else int(Lmax_batch)
        n_s = len(tiles)
        text = None
        if Lmax > 0:
            kind = torch.full((n_s, Lmax), 2, dtype=torch.int32)
            idx = torch.zeros((n_s, Lmax), dtype=torch.int32)
            for i, t in enumerate(text_ids):
                kind[i, :t.numel()] = 0
                idx[i, :t.numel()] = t.to(torch.int32)
            text = eng.splice(kind.view(-1), idx.view(-1), None).view(n_s, Lmax, -1)
        tile_sample = [s for s, n in enumerate(n_tiles) for _ in range(n)]
        T = len(tile_sample)
        feat_row0 = [0]
        for n in n_tiles:
            feat_row0.append(feat_row0[-1] + n * nq)
        rows_kind, rows_idx, rows_lab = [], [], []
        img_i = 0
        for b in range(Bsz):
            ids = ids_cpu[b][mask_cpu[b]]
            lab = lab_cpu[b][mask_cpu[b]]
            is_img = ids == IMAGE_TOKEN_INDEX
            n_img = int(is_img.sum())
            if n_img == 0:
                rows_kind.append(torch.zeros(ids.numel(), dtype=torch.int32)); rows_idx.append(ids.to(torch.int32)); rows_lab.append(lab)
                img_i += 1
                continue
            kp, ip, lp = [], [], []
            cuts = [-1] + torch.where(is_img)[0].tolist() + [ids.numel()]
            for j in range(len(cuts) - 1):
                seg = ids[cuts[j] + 1:cuts[j + 1]]
                kp.append(torch.zeros(seg.numel(), dtype=torch.int32)); ip.append(seg.to(torch.int32)); lp.append(lab[cuts[j] + 1:cuts[j + 1]])
                if j < n_img:
                    n = n_tiles[img_i] * nq
                    kp.append(torch.ones(n, dtype=torch.int32))
                    ip.append(torch.arange(feat_row0[img_i], feat_row0[img_i] + n, dtype=torch.int32))
                    lp.append(torch.full((n,), IGNORE_INDEX, dtype=lab.dtype))
                    img_i += 1
            rows_kind.append(torch.cat(kp)); rows_idx.append(torch.cat(ip)); rows_lab.append(torch.cat(lp))
        max_len = getattr(self.model.config, "tokenizer_model_max_length", None)
        if max_len is not None:
            rows_kind, rows_idx, rows_lab = ([r[:max_len] for r in x] for x in (rows_kind, rows_idx, rows_lab))
        S = max(int(r.numel()) for r in rows_kind)
        kind = torch.full((Bsz, S), 2, dtype=torch.int32)
        idx = torch.zeros((Bsz, S), dtype=torch.int32)
        lab_out = torch.full((Bsz, S), IGNORE_INDEX, dtype=torch.int32)
        pos = torch.zeros((Bsz, S), dtype=torch.int32)
        seqlens = []
        for b in range(Bsz):
            n = int(rows_kind[b].numel())
            kind[b, :n] = rows_kind[b]; idx[b, :n] = rows_idx[b]; lab_out[b, :n] = rows_lab[b].to(torch.int32); pos[b, :n] = torch.arange(n, dtype=torch.int32)
            seqlens.append(max(1, n))
        vis_rows = torch.full((T * nq,), -1, dtype=torch.int32)
        flat_kind, flat_idx = kind.view(-1), idx.view(-1)
        sel = torch.nonzero(flat_kind == 1).view(-1)
        vis_rows[flat_idx[sel].long()] = sel.to(torch.int32)
        images = torch.cat([t.to(dev, torch.bfloat16) for t in tiles], 0).contiguous()
        d = lambda t: t.to(dev).contiguous()             # noqa: E731
        kind_d, idx_d, vis_d, pos_d, lab_d = d(kind.view(-1)), d(idx.view(-1)), d(vis_rows), d(pos), d(lab_out)
        ts = (C.c_int * T)(*tile_sample)
        sl = (C.c_int * Bsz)(*seqlens)
        B.check(self.lib.vz_train_stage1_accumulate(self.h, B.ptr(images), T, B.ptr(text), n_s, Lmax, ts, B.ptr(kind_d), B.ptr(idx_d), B.ptr(vis_d),
                                                    Bsz, S, sl, B.ptr(pos_d), B.ptr(lab_d), float(inv_n), eng._s()))
        out = C.c_double(0.0)
        B.check(self.lib.vz_train_loss_sum(self.h, C.byref(out), eng._s()))
        return out.value

    def forward_backward(self, input_ids, attention_mask, labels, images, micro_batch: Optional[int] = None) -> float:
        """loss (mean token cross-entropy over the batch's valid targets) with the gradients of the projector left in the arena."""
        ids_cpu = input_ids.detach().to("cpu", torch.long)
        Bsz = ids_cpu.shape[0]
        mask_cpu = torch.ones_like(ids_cpu, dtype=torch.bool) if attention_mask is None else attention_mask.detach().to("cpu").bool()
        lab_cpu = labels.detach().to("cpu", torch.long)
        tiles = [x.unsqueeze(0) if x.ndim == 3 else x for x in (images if isinstance(images, (list, tuple)) else [images[i] for i in range(images.shape[0])])]
        if len(tiles) != Bsz:
            raise ValueError("one image (tile stack) per sample expected in a Stage-1 batch")
        # valid targets of the WHOLE batch: labels shifted by one inside each spliced row; image positions are IGNORE, so only text labels count
        n_valid = 0
        for b in range(Bsz):
            lab = lab_cpu[b][mask_cpu[b]]
            ids = ids_cpu[b][mask_cpu[b]]
            spliced = []
            for t, l in zip(ids.tolist(), lab.tolist()):
                spliced.append(l if t != IMAGE_TOKEN_INDEX else None)
            flat = []
            for v in spliced:
                flat.extend([IGNORE_INDEX] * (tiles[b].shape[0] * self.eng.cfg.qf_queries) if v is None else [v])
            max_len = getattr(self.model.config, "tokenizer_model_max_length", None)
            flat = flat if max_len is None else flat[:max_len]
            n_valid += sum(1 for v in flat[1:] if v != IGNORE_INDEX)
        if n_valid == 0:
            raise ValueError("no valid target in the batch")
        mb = Bsz if micro_batch is None else max(1, int(micro_batch))
        Lmax_batch = max(int((ids_cpu[b] != IMAGE_TOKEN_INDEX).sum()) for b in range(Bsz))
        total = 0.0
        for b0 in range(0, Bsz, mb):
            sl = slice(b0, min(Bsz, b0 + mb))
            total += self._accumulate(ids_cpu[sl], mask_cpu[sl], lab_cpu[sl], tiles[sl], 1.0 / n_valid, Lmax_batch)
        return total / n_valid

    def all_reduce(self):
        B.check(self.lib.vz_train_allreduce(self.h, self.eng._s()))

    def optimizer_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        B.check(self.lib.vz_train_adamw_step(self.h, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), self.eng._s()))
        self.steps_done += 1

    def step(self, input_ids, attention_mask, labels, images, lr: Optional[float] = None, total_steps: Optional[int] = None,
             micro_batch: Optional[int] = None) -> float:
        """one optimiser step as HF's Trainer takes it: forward + backward, (all-reduce: mean over the data-parallel ranks), AdamW with
        the scheduled learning rate.  Either `lr` (explicit) or `total_steps` (the schedule's length: the Trainer knows it from the
        dataset and the epoch count; there is no sensible default) must be given."""
        if lr is None and total_steps is None:
            raise ValueError("Stage1Trainer.step: pass lr=..., or total_steps=... for the cosine schedule around mm_projector_lr = 2e-3")
        loss = self.forward_backward(input_ids, attention_mask, labels, images, micro_batch)
        self.all_reduce()
        self.optimizer_step(lr_at(self.steps_done, int(total_steps)) if lr is None else lr)
        return loss

    def projector_state_dict(self) -> Dict[str, torch.Tensor]:
        """the fp32 masters under the reference's checkpoint keys (`model.mm_projector.*`): k_proj / v_proj split out
        of the engine's stacked cross-attention matrix, the cross-attention `in_proj_bias` = q | k | v concatenated."""
        out = {}
        for ref, (eng_name, rows) in _ref_to_engine(self.eng.cfg).items():
            if isinstance(eng_name, tuple):
                t = torch.cat([self.master(n).reshape(-1) for n in eng_name])
            else:
                t = self.master(eng_name)
                t = t if rows is None else t[rows]
            out["model.mm_projector." + ref] = t.detach().to("cpu", torch.float32).clone()
        return out

    def save_projector(self, path: str, dtype: torch.dtype = torch.bfloat16) -> str:
        """write the reference's Stage-1 checkpoint: `mm_projector.bin`, a flat torch.save dict of the projector's parameters under
        `model.mm_projector.*` keys (ref:vis_zephyr/train/vis_zephyr_trainer.py:304-348 saves `named_parameters()` filtered by
        'mm_projector' in the training dtype, bf16; ref:vis_zephyr/model/builder.py:118-120 loads it with strict=False) - the file
        `load_pretrained_model` of either implementation ingests.  `path` = the file, or a directory that receives mm_projector.bin."""
        import os
        if os.path.isdir(path):
            path = os.path.join(path, "mm_projector.bin")
        torch.save({k: v.to(dtype) for k, v in self.projector_state_dict().items()}, path)
        return path

    def init_comm(self):
        """data-parallel communicator (RCCL) over torch.distributed's ranks: rank 0's unique id travels over the process group."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        buf = C.create_string_buffer(128)
        if dist.get_rank() == 0:
            B.check(self.lib.vz_comm_unique_id(buf))
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0)
        B.check(self.lib.vz_train_comm_init(self.h, box[0], dist.get_rank(), dist.get_world_size()))

    def init_comm_single_rank(self):
        buf = C.create_string_buffer(128)
        B.check(self.lib.vz_comm_unique_id(buf))
        B.check(self.lib.vz_train_comm_init(self.h, bytes(buf.raw), 0, 1))


def _from_ptr(ptr: int, n: int, device) -> torch.Tensor:
    """fp32 view of `n` floats of device memory the trainer owns (no copy, no ownership): through the CUDA array interface"""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device=device)
