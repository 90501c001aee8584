"""Python owner of one native engine (vz_engine): weight packing from the reference's state-dict
names into the engine's HBM layout, and thin stage calls.  torch is plumbing only (device memory,
dtype casts at load time, streams); every stage runs in libviszephyr_hip.so.

HBM layout: all matrices bf16 row-major [out_features, in_features] exactly
as the reference stores them, except
  * q/k/v (CLIP, Zephyr) stacked into one [3C,C] / [6144,4096] matrix -> one GEMM per layer,
  * cross-attention k|v stacked [8192,5120],
  * Zephyr gate/up interleaved in 16-row groups [16 gate | 16 up | ...] so the SwiGLU product is
    formed inside the GEMM epilogue,
  * the CLIP patch convolution flattened to [1024, 588] and zero-padded to K = 640.
Vectors (biases, LayerNorm/RMSNorm scales) are fp32.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import binding as B
from . import tp as TP
from .synth import ArchConfig, iter_state_dict

_CLIP_PREFIXES = ("model.vision_tower.vision_tower.vision_model.", "model.vision_tower.vision_tower.")
_QF = "model.mm_projector."


def rope_tables(cfg: ArchConfig, max_pos: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 cos/sin [max_pos, head_dim/2] (hf:models/mistral/modeling_mistral.py:262-317: inv_freq =
    theta^(-2i/d), angle = pos * inv_freq computed in fp32)."""
    d = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    fr = torch.arange(max_pos, dtype=torch.float32).unsqueeze(-1) * inv
    return fr.cos().contiguous(), fr.sin().contiguous()


class Engine:
    def __init__(self, cfg: ArchConfig, device="cuda:0", max_batch: int = 1, max_ctx: int = 4096,
                 max_tiles: int = 8, max_text: int = 2048, tp_size: int = 1, tp_rank: int = 0, weight_fp8: bool = False,
                 weight_nf4: bool = False):
        self.lib = B.load_library()            # raises when the HIP library is absent: no fallback
        self._qf_kv_pool = {}
        if not torch.cuda.is_available():
            raise RuntimeError("vz_hip.Engine needs a ROCm GPU (gfx950); there is no CPU fallback")
        self.cfg = cfg
        TP.check_divisible(cfg, tp_size)
        self.tp_size, self.tp_rank = tp_size, tp_rank
        self.vp = (cfg.vocab + tp_size - 1) // tp_size          # lm_head rows per rank (zero padded)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.max_batch, self.max_ctx, self.max_tiles, self.max_text = max_batch, max_ctx, max_tiles, max_text
        c = B.VzConfig(
            hidden=cfg.hidden, inter=cfg.inter, n_layers=cfg.n_layers, n_heads=cfg.n_heads, n_kv_heads=cfg.n_kv_heads,
            head_dim=cfg.head_dim, vocab=cfg.vocab, rms_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
            sliding_window=cfg.sliding_window, clip_hidden=cfg.clip_hidden, clip_inter=cfg.clip_inter,
            clip_layers=cfg.clip_layers, clip_heads=cfg.clip_heads, clip_image=cfg.clip_image,
            clip_patch=cfg.clip_patch, clip_eps=cfg.clip_eps, qf_queries=cfg.qf_queries, qf_blocks=cfg.qf_blocks,
            qf_heads=cfg.qf_heads, qf_kv_dim=cfg.qf_kv_dim, qf_eps=cfg.qf_eps, fusion_groups=cfg.fusion_groups,
            fusion_layers_per_group=cfg.fusion_layers_per_group, max_batch=max_batch, max_ctx=max_ctx,
            max_tiles=max_tiles, max_text=max_text, tp_size=tp_size, tp_rank=tp_rank,
            clip_keep_cls=int(cfg.clip_keep_cls), weight_fp8=int(weight_fp8))
        self.weight_fp8 = bool(weight_fp8)
        if weight_fp8 and weight_nf4:
            raise ValueError("weight_fp8 and weight_nf4 are two quantisations of the same linears: pick one")
        self.weight_nf4 = bool(weight_nf4)
        self._nf4_done = set()
        self.prefill_fp8 = False
        h = C.c_void_p()
        B.check(self.lib.vz_engine_create(C.byref(c), C.byref(h)))
        self.h = h
        self.w: Dict[str, torch.Tensor] = {}
        self._registered = set()
        cos, sin = rope_tables(cfg, max_ctx)
        self._cos, self._sin = cos.to(self.device), sin.to(self.device)
        B.check(self.lib.vz_engine_set_rope(self.h, B.ptr(self._cos), B.ptr(self._sin), max_ctx))
        self.ready = False

    def set_prefill_fp8(self, on: bool = True):
        """weight_fp8 engines: run the Zephyr prefill linears on the fp8 MFMA (activations quantised to e4m3 per row on the device,
        weights = the e4m3 copies the decode stream reads) instead of bf16 MFMA on the dequantised tensors.  Off by default."""
        B.check(self.lib.vz_engine_prefill_fp8(self.h, int(bool(on))))
        self.prefill_fp8 = bool(on)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.vz_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --------------------------------------------------------------------------------------------
    # weights
    # --------------------------------------------------------------------------------------------
    @staticmethod
    def _fp8_names(name: str):
        return (name + "8", name + "s") if name.endswith("lm_head") else (name + "8", name[:-1] + "ws")

    def _drop_fp8_copy(self, name: str):
        """the e4m3 copy + row scales of a bf16 decode weight are derived data: whoever rewrites (or replaces) the bf16 tensor
        invalidates them, so that finalize() re-quantises (a second load_weights / LoRA re-merge / resize_vocab on a
        weight_fp8 engine must not leave decode streaming the OLD e4m3 rows)."""
        self._nf4_done.discard(name)          # a rewritten tensor holds unquantised values again
        nt = name + "t"
        if nt in self.w:                      # the fragment-tiled copy of a bf16 decode weight (bf16 engines)
            del self.w[nt]
            self._registered.discard(nt)
            B.check(self.lib.vz_engine_unset_weight(self.h, nt.encode()))
        if not (self.weight_fp8 and self._FP8_NAMES.match(name)):
            return
        n8, ns = self._fp8_names(name)
        for n in (n8, ns, n8 + "t"):          # e4m3 rows, their scales, the fragment-tiled e4m3 copy
            if n in self.w:
                del self.w[n]
                self._registered.discard(n)
                B.check(self.lib.vz_engine_unset_weight(self.h, n.encode()))

    _QF_KV = re.compile(r"qf\.(\d+)\.ca_kv\.(w|b)$")

    def _dest(self, name: str, shape, dtype) -> torch.Tensor:
        self._drop_fp8_copy(name)
        t = self.w.get(name)
        if t is None:
            m = self._QF_KV.match(name)
            if m:
                # the cross-attention K|V projections of all Q-Former blocks read the same pre-normed visual features: laid out back
                # to back (block-major), vz_qformer runs them as ONE GEMM over N = blocks * 2H (whole 256^2 tiles, no stream-K tail)
                pool = self._qf_kv_pool.get(m.group(2))
                if pool is None:
                    pool = torch.zeros(self.cfg.qf_blocks, *shape, dtype=dtype, device=self.device)
                    self._qf_kv_pool[m.group(2)] = pool
                t = pool[int(m.group(1))]
            else:
                t = torch.zeros(*shape, dtype=dtype, device=self.device)
            self.w[name] = t
        return t

    def _mat(self, name, shape, src: torch.Tensor, rows: Optional[slice] = None):
        d = self._dest(name, shape, torch.bfloat16)
        s = src.to(self.device, non_blocking=True).to(torch.bfloat16)
        (d if rows is None else d[rows]).copy_(s.reshape((d if rows is None else d[rows]).shape))

    def _vec(self, name, n, src: torch.Tensor, sl: Optional[slice] = None):
        d = self._dest(name, (n,), torch.float32)
        s = src.to(self.device, non_blocking=True).to(torch.float32)
        (d if sl is None else d[sl]).copy_(s.reshape(-1))

    def add_weight(self, name: str, t: torch.Tensor) -> bool:
        """Consume one tensor under its reference state-dict key.  Returns False for keys that are
        not on the hot path (CLIP post_layernorm, buffers)."""
        cfg = self.cfg
        H, Cc, I = cfg.hidden, cfg.clip_hidden, cfg.inter
        for pre in _CLIP_PREFIXES:
            if name.startswith(pre):
                return self._add_clip(name[len(pre):], t)
        if name.startswith(_QF):
            return self._add_qformer(name[len(_QF):], t)
        t = TP.shard(name, t, self.tp_rank, self.tp_size)       # this rank's slice (identity at tp_size 1 / replicated tensors)
        I = cfg.inter // self.tp_size
        if name == "model.embed_tokens.weight":
            self._mat("llm.embed", (cfg.vocab, H), t)
        elif name == "lm_head.weight":
            self._mat("llm.lm_head", (self.vp, H), t, slice(0, t.shape[0]))
        elif name == "model.norm.weight":
            self._vec("llm.norm", H, t)
        else:
            m = re.fullmatch(r"model\.layers\.(\d+)\.(.+)", name)
            if not m:
                return False
            i, rest = int(m.group(1)), m.group(2)
            p = f"llm.{i}."
            qd, kvd = cfg.n_heads * cfg.head_dim // self.tp_size, cfg.n_kv_heads * cfg.head_dim // self.tp_size
            if rest == "input_layernorm.weight":
                self._vec(p + "in_norm", H, t)
            elif rest == "post_attention_layernorm.weight":
                self._vec(p + "post_norm", H, t)
            elif rest == "self_attn.q_proj.weight":
                self._mat(p + "qkv.w", (qd + 2 * kvd, H), t, slice(0, qd))
            elif rest == "self_attn.k_proj.weight":
                self._mat(p + "qkv.w", (qd + 2 * kvd, H), t, slice(qd, qd + kvd))
            elif rest == "self_attn.v_proj.weight":
                self._mat(p + "qkv.w", (qd + 2 * kvd, H), t, slice(qd + kvd, qd + 2 * kvd))
            elif rest == "self_attn.o_proj.weight":
                self._mat(p + "o.w", (H, qd), t)
            elif rest in ("mlp.gate_proj.weight", "mlp.up_proj.weight"):
                d = self._dest(p + "gu.w", (2 * I, H), torch.bfloat16).view(I // 16, 2, 16, H)
                d[:, 0 if "gate" in rest else 1].copy_(t.to(self.device).to(torch.bfloat16).view(I // 16, 16, H))
            elif rest == "mlp.down_proj.weight":
                self._mat(p + "down.w", (H, I), t)
            else:
                return False
        return True

    def _add_clip(self, k: str, t: torch.Tensor) -> bool:
        cfg = self.cfg
        Cc = cfg.clip_hidden
        kreal = 3 * cfg.clip_patch * cfg.clip_patch
        kpad = (kreal + 63) // 64 * 64
        if k == "embeddings.class_embedding":
            self._mat("clip.cls", (Cc,), t)
        elif k == "embeddings.patch_embedding.weight":
            d = self._dest("clip.patch_w", (Cc, kpad), torch.bfloat16)
            d[:, :kreal].copy_(t.to(self.device).to(torch.bfloat16).reshape(Cc, kreal))
        elif k == "embeddings.position_embedding.weight":
            self._mat("clip.pos", (cfg.clip_tokens, Cc), t)
        elif k.startswith("pre_layrnorm."):
            self._vec("clip.pre_ln." + ("w" if k.endswith("weight") else "b"), Cc, t)
        else:
            m = re.fullmatch(r"encoder\.layers\.(\d+)\.(.+)\.(weight|bias)", k)
            if not m:
                return False
            i, mod, wb = int(m.group(1)), m.group(2), m.group(3)
            p = f"clip.{i}."
            isw = wb == "weight"
            if mod in ("layer_norm1", "layer_norm2"):
                self._vec(p + ("ln1." if mod.endswith("1") else "ln2.") + ("w" if isw else "b"), Cc, t)
            elif mod in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"):
                j = "qkv".index(mod[-6])
                sl = slice(j * Cc, (j + 1) * Cc)
                if isw:
                    self._mat(p + "qkv.w", (3 * Cc, Cc), t, sl)
                else:
                    self._vec(p + "qkv.b", 3 * Cc, t, sl)
            elif mod == "self_attn.out_proj":
                self._mat(p + "o.w", (Cc, Cc), t) if isw else self._vec(p + "o.b", Cc, t)
            elif mod == "mlp.fc1":
                self._mat(p + "fc1.w", (cfg.clip_inter, Cc), t) if isw else self._vec(p + "fc1.b", cfg.clip_inter, t)
            elif mod == "mlp.fc2":
                self._mat(p + "fc2.w", (Cc, cfg.clip_inter), t) if isw else self._vec(p + "fc2.b", Cc, t)
            else:
                return False
        return True

    def _add_qformer(self, k: str, t: torch.Tensor) -> bool:
        cfg = self.cfg
        H, KD = cfg.hidden, cfg.qf_kv_dim
        if k == "learned_queries":
            self._mat("qf.queries", (cfg.qf_queries, H), t)
        elif k in ("pre_norm.weight", "pre_norm.bias"):
            self._vec("qf.pre_norm." + ("w" if k.endswith("weight") else "b"), KD, t)
        elif k in ("norm.weight", "norm.bias"):
            self._vec("qf.norm." + ("w" if k.endswith("weight") else "b"), H, t)
        else:
            m = re.fullmatch(r"blocks\.(\d+)\.(.+)", k)
            if not m:
                return False
            i, rest = int(m.group(1)), m.group(2)
            p = f"qf.{i}."
            mm = re.fullmatch(r"norm([123])\.(weight|bias)", rest)
            if mm:
                self._vec(p + f"n{mm.group(1)}." + ("w" if mm.group(2) == "weight" else "b"), H, t)
            elif rest == "self_attn.in_proj_weight":
                self._mat(p + "sa_in.w", (3 * H, H), t)
            elif rest == "self_attn.in_proj_bias":
                self._vec(p + "sa_in.b", 3 * H, t)
            elif rest == "self_attn.out_proj.weight":
                self._mat(p + "sa_out.w", (H, H), t)
            elif rest == "self_attn.out_proj.bias":
                self._vec(p + "sa_out.b", H, t)
            elif rest == "cross_attn.q_proj_weight":
                self._mat(p + "ca_q.w", (H, H), t)
            elif rest == "cross_attn.k_proj_weight":
                self._mat(p + "ca_kv.w", (2 * H, KD), t, slice(0, H))
            elif rest == "cross_attn.v_proj_weight":
                self._mat(p + "ca_kv.w", (2 * H, KD), t, slice(H, 2 * H))
            elif rest == "cross_attn.in_proj_bias":
                tt = t.reshape(-1)
                self._vec(p + "ca_q.b", H, tt[:H])
                self._vec(p + "ca_kv.b", 2 * H, tt[H:])
            elif rest == "cross_attn.out_proj.weight":
                self._mat(p + "ca_out.w", (H, H), t)
            elif rest == "cross_attn.out_proj.bias":
                self._vec(p + "ca_out.b", H, t)
            elif rest == "ffn.0.weight":
                self._mat(p + "ffn1.w", (2 * H, H), t)
            elif rest == "ffn.0.bias":
                self._vec(p + "ffn1.b", 2 * H, t)
            elif rest == "ffn.2.weight":
                self._mat(p + "ffn2.w", (H, 2 * H), t)
            elif rest == "ffn.2.bias":
                self._vec(p + "ffn2.b", H, t)
            else:
                return False
        return True

    def load_weights(self, named: Iterable[Tuple[str, torch.Tensor]]):
        for name, t in named:
            self.add_weight(name, t)
        self.finalize()

    def init_comm(self):
        """create the RCCL communicator of a tensor-parallel engine: rank 0's unique id travels over torch.distributed
        (already initialised by the launcher), then every rank joins."""
        if self.tp_size == 1:
            return
        import torch.distributed as dist
        buf = C.create_string_buffer(128)
        if self.tp_rank == 0:
            B.check(self.lib.vz_comm_unique_id(buf))
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0)
        B.check(self.lib.vz_comm_init(self.h, box[0]))
        if os.environ.get("VZ_TP_ONESHOT", "0") == "1":
            self.init_oneshot()

    def init_oneshot(self) -> bool:
        """Opt-in (VZ_TP_ONESHOT=1): the decode step's [B, hidden] all-reduces on the one-shot kernel (csrc/comm_oneshot.hip) instead of
        RCCL - every rank's receive area is exported with hipIpcGetMemHandle, the handles travel over torch.distributed, the peers' areas
        are opened with hipIpcOpenMemHandle (peer-mapped over xGMI) and handed to the engine in rank order.  Returns False (RCCL stays)
        if any rank fails to map a peer.  UNMEASURED: no multi-GPU box has been available to the build; the kernel's protocol and
        arithmetic are tested in one process (tests/test_oneshot_gpu.py), this wiring only by tests/test_tp_gloo.py's dry plan."""
        import torch.distributed as dist
        class IpcHandle(C.Structure):                            # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
            _fields_ = [("reserved", C.c_char * 64)]
        hip = C.CDLL("libamdhip64.so")
        hip.hipIpcGetMemHandle.argtypes = [C.POINTER(IpcHandle), C.c_void_p]
        hip.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), IpcHandle, C.c_uint]
        area, nbytes = C.c_void_p(), C.c_size_t()
        B.check(self.lib.vz_comm_oneshot_local(self.h, C.byref(area), C.byref(nbytes)))
        handle = IpcHandle()
        ok = hip.hipIpcGetMemHandle(C.byref(handle), area) == 0
        box = [None] * self.tp_size
        dist.all_gather_object(box, (ok, bytes(C.string_at(C.byref(handle), 64))))
        areas = (C.c_void_p * self.tp_size)()
        good = all(o for o, _ in box)
        if good:
            for q, (_, h) in enumerate(box):
                if q == self.tp_rank:
                    areas[q] = area.value
                    continue
                peer = C.c_void_p()
                hq = IpcHandle()
                C.memmove(C.byref(hq), h, 64)
                if hip.hipIpcOpenMemHandle(C.byref(peer), hq, 1) != 0:        # hipIpcMemLazyEnablePeerAccess
                    good = False
                    break
                areas[q] = peer.value
        flags = [None] * self.tp_size
        dist.all_gather_object(flags, bool(good))
        if not all(flags):
            return False
        B.check(self.lib.vz_comm_oneshot_attach(self.h, areas, self.tp_size))
        self.oneshot = True
        return True

    def resize_vocab(self, n: int):
        """grow (new rows = mean of the old ones, as vz_hip.weights.resize_vocab does at load time) or shrink the
        embedding and lm_head tables of a live engine (HF `resize_token_embeddings`; ref builder.py:141-153)."""
        import dataclasses
        if self.tp_size != 1:
            raise NotImplementedError("resize_vocab on a tensor-parallel engine: rebuild it at the new vocabulary")
        old = self.cfg.vocab
        if n == old:
            return
        tables = {}
        for name in ("llm.embed", "llm.lm_head"):
            t = self.w[name]
            if n > old:
                extra = t.float().mean(0, keepdim=True).to(t.dtype).expand(n - old, -1)
                tables[name] = torch.cat([t, extra], 0).contiguous()
            else:
                tables[name] = t[:n].contiguous()
        torch.cuda.synchronize(self.device)
        B.check(self.lib.vz_engine_resize_vocab(self.h, n))
        self.cfg = dataclasses.replace(self.cfg, vocab=n)
        self.vp = n
        for name, t in tables.items():
            self._drop_fp8_copy(name)          # old-size e4m3 table / scales of lm_head
            self.w[name] = t
            self._registered.discard(name)
        self.finalize()

    def init_comm_single_rank(self):
        """self-test: give a tp_size == 1 engine a ONE-rank RCCL communicator; with vz_tune_set(7, 1) its all-reduce /
        all-gather call sites then really go through RCCL (identity results) on a single GPU."""
        assert self.tp_size == 1
        buf = C.create_string_buffer(128)
        B.check(self.lib.vz_comm_unique_id(buf))
        B.check(self.lib.vz_comm_init(self.h, bytes(buf.raw)))

    def all_gather(self, send: torch.Tensor) -> torch.Tensor:
        """RCCL all-gather of one equally sized contiguous tensor per rank -> [tp_size, *send.shape] (rank order)."""
        send = send.to(self.device).contiguous()
        recv = torch.empty((self.tp_size,) + tuple(send.shape), dtype=send.dtype, device=self.device)
        B.check(self.lib.vz_tp_all_gather(self.h, B.ptr(send), B.ptr(recv), send.numel() * send.element_size(), self._s()))
        return recv

    def load_synthetic(self, seed: int = 0):
        """hash-generated weights, produced on the device (bit-identical to the CPU generator)."""
        self.load_weights(iter_state_dict(self.cfg, seed, device=self.device))

    _FP8_NAMES = re.compile(r"llm\.(\d+\.(qkv|o|gu|down)\.w|lm_head)$")

    def _quantize_decode_weights(self):
        """weight_fp8: every decode-side Zephyr linear gets an e4m3 copy + per-row 2^e scales (vz_hip/quant.py); the bf16
        tensor is replaced by the exactly-equal dequantised values, so prefill (bf16 MFMA) and decode (fp8 stream) agree."""
        from . import quant
        for name in [n for n in self.w if self._FP8_NAMES.match(n)]:
            n8, ns = self._fp8_names(name)
            if n8 in self.w:
                continue
            w = self.w[name]
            w8, scale = quant.quantize_rows(w)
            w.copy_(quant.dequantize_rows(w8, scale).to(torch.bfloat16))
            self.w[n8], self.w[ns] = w8.contiguous(), scale.contiguous()
            self._registered.discard(name)

    def _tile_decode_weights(self):
        """every decode-side Zephyr linear gets a copy in MFMA-fragment order that the 2..64-row decode steps stream with 1-KiB
        wave-instructions; same values, same k order as the row-major tensor the prefill GEMMs and the 1-row GEMV keep using.
        bf16 engines: vz_op_tile_weights on the bf16 tensors (+14.5 GB for Zephyr-7B).  e4m3 engines (round 3): vz_op_tile_weights_fp8
        on the e4m3 copies (+7.3 GB; their 17..64-row steps stream those, 2..16 rows keep the row-major e4m3 rows).
        VZ_DECODE_TILED=0 turns it off."""
        if os.environ.get("VZ_DECODE_TILED", "1") == "0":
            return
        for name in [n for n in self.w if self._FP8_NAMES.match(n)]:
            if self.weight_fp8:
                n8 = self._fp8_names(name)[0]
                w8 = self.w.get(n8)
                if w8 is None or n8 + "t" in self.w or w8.shape[0] % 128 or w8.shape[1] % 1024:
                    continue
                self.w[n8 + "t"] = B.tile_weights_fp8(w8)
                continue
            w = self.w[name]
            if name + "t" in self.w or w.shape[0] % 16 or w.shape[1] % 64:
                continue
            self.w[name + "t"] = B.tile_weights(w)

    _NF4_NAMES = re.compile(r"llm\.\d+\.(qkv|o|gu|down)\.w$")       # the decoder layers' linears; lm_head stays as it is (bitsandbytes' default skip list)

    def _nf4_decode_weights(self):
        """weight_nf4 (`load_4bit`): every decoder-layer linear is replaced by its NF4 fake-quantisation (vz_hip/quant.py: 64-element
        blocks, fp32 absmax) rounded to bf16; every kernel then runs on those values - the 4-bit model on the bf16 engine."""
        from . import quant
        for name in [n for n in self.w if self._NF4_NAMES.match(n) and n not in self._nf4_done]:
            w = self.w[name]
            w.copy_(quant.fake_quantize_nf4(w).to(torch.bfloat16))
            self._nf4_done.add(name)
            self._registered.discard(name)

    def finalize(self):
        if self.weight_nf4:
            self._nf4_decode_weights()
        if self.weight_fp8:
            self._quantize_decode_weights()
        if self.max_batch > (16 if self.weight_fp8 else 1):      # (an e4m3 engine streams row-major e4m3 rows up to 16 rows, the tiled e4m3 copies beyond)
            self._tile_decode_weights()
        for name, t in self.w.items():
            if name in self._registered:
                continue
            B.check(self.lib.vz_engine_set_weight(self.h, name.encode(), B.ptr(t),
                                                  0 if t.dtype == torch.bfloat16 else (2 if t.dtype == torch.uint8 else 1),
                                                  t.numel()))
            self._registered.add(name)
        B.check(self.lib.vz_engine_finalize(self.h))
        self.ready = True

    # --------------------------------------------------------------------------------------------
    # stages
    # --------------------------------------------------------------------------------------------
    def _s(self):
        return B.stream_ptr(self.device)

    def clip_fused_features(self, images: torch.Tensor, return_hidden: bool = False):
        """images [T,3,336,336] -> bf16 [T,576,5*C]  (+ all hidden states [L+1,T,577,C] for tests)."""
        cfg = self.cfg
        if images.dim() != 4 or images.shape[1] != 3 or images.shape[2] != cfg.clip_image or images.shape[3] != cfg.clip_image:
            raise ValueError(f"Input image size ({images.shape[-2]}*{images.shape[-1]}) doesn't match model "
                             f"({cfg.clip_image}*{cfg.clip_image}).")
        x = images.to(self.device, torch.bfloat16).contiguous()
        T = x.shape[0]
        out = torch.empty(T, cfg.vision_tokens, (cfg.fusion_groups + 1) * cfg.clip_hidden, dtype=torch.bfloat16,
                          device=self.device)
        hid = None
        if return_hidden:
            hid = torch.empty(cfg.clip_layers + 1, T, cfg.clip_tokens, cfg.clip_hidden, dtype=torch.bfloat16,
                              device=self.device)
        B.check(self.lib.vz_clip_fused_features(self.h, B.ptr(x), T, B.ptr(out), B.ptr(hid), self._s()))
        return (out, hid) if return_hidden else out

    def qformer(self, feats: torch.Tensor, text: Optional[torch.Tensor], tile_sample: Sequence[int]):
        """feats bf16 [T,576,5120]; text bf16 [n_samples,Lmax,H] or None; tile_sample[t] = sample of tile t."""
        cfg = self.cfg
        feats = feats.to(self.device, torch.bfloat16).contiguous()
        T = feats.shape[0]
        if text is not None and text.shape[1] == 0:
            text = None
        if text is not None:
            text = text.to(self.device, torch.bfloat16).contiguous()
            n_samples, Lmax = text.shape[0], text.shape[1]
        else:
            n_samples, Lmax = (max(tile_sample) + 1 if len(tile_sample) else 1), 0
        ts = (C.c_int * T)(*[int(v) for v in tile_sample])
        out = torch.empty(T, cfg.qf_queries, cfg.hidden, dtype=torch.bfloat16, device=self.device)
        B.check(self.lib.vz_qformer(self.h, B.ptr(feats), T, B.ptr(text), n_samples, Lmax, ts, B.ptr(out), self._s()))
        return out

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        ids32 = ids.to(self.device, torch.int32).contiguous().view(-1)
        if ids32.numel() == 0:
            return torch.empty(*ids.shape, self.cfg.hidden, dtype=torch.bfloat16, device=self.device)
        if int(ids32.min()) < 0 or int(ids32.max()) >= self.cfg.vocab:
            raise IndexError("token id out of range for embed_tokens")
        out = torch.empty(ids32.numel(), self.cfg.hidden, dtype=torch.bfloat16, device=self.device)
        B.check(self.lib.vz_embed_splice(self.h, None, B.ptr(ids32), ids32.numel(), None, B.ptr(out), self._s()))
        return out.view(*ids.shape, self.cfg.hidden)

    def splice(self, kind: torch.Tensor, idx: torch.Tensor, visual: Optional[torch.Tensor]) -> torch.Tensor:
        """rows: kind 0 = embedding row idx, 1 = visual row idx, 2 = zero."""
        kind = kind.to(self.device, torch.int32).contiguous()
        idx = idx.to(self.device, torch.int32).contiguous()
        out = torch.empty(kind.numel(), self.cfg.hidden, dtype=torch.bfloat16, device=self.device)
        vis = None if visual is None else visual.to(torch.bfloat16).contiguous()
        B.check(self.lib.vz_embed_splice(self.h, B.ptr(kind), B.ptr(idx), kind.numel(), B.ptr(vis), B.ptr(out), self._s()))
        return out

    def prefill(self, embeds: torch.Tensor, seqlens: Sequence[int], position_ids: Optional[torch.Tensor] = None,
                all_logits: bool = False, last_logits: bool = True):
        """embeds bf16 [B,S,H] right-padded.  Returns (logits_all fp32 [B,S,V] | None, logits_last fp32 [B,V] | None)."""
        cfg = self.cfg
        x = embeds.to(self.device, torch.bfloat16).contiguous()
        Bn, S = x.shape[0], x.shape[1]
        if position_ids is None:
            position_ids = torch.arange(S, dtype=torch.int32, device=self.device).unsqueeze(0).expand(Bn, S)
        pos = position_ids.to(self.device, torch.int32).contiguous()
        la = torch.empty(Bn, S, cfg.vocab, dtype=torch.float32, device=self.device) if all_logits else None
        ll = torch.empty(Bn, cfg.vocab, dtype=torch.float32, device=self.device) if last_logits else None
        sl = (C.c_int * Bn)(*[int(v) for v in seqlens])
        B.check(self.lib.vz_llm_prefill(self.h, B.ptr(x), Bn, S, sl, B.ptr(pos), B.ptr(la), B.ptr(ll), self._s()))
        return la, ll

    def prefill_rows(self, row0: int, embeds: torch.Tensor, seqlens: Sequence[int], position_ids: Optional[torch.Tensor] = None):
        """`prefill` into KV-cache rows row0 .. row0+B-1 (continuous batching); returns the last-position logits fp32 [B,V]."""
        x = embeds.to(self.device, torch.bfloat16).contiguous()
        Bn, S = x.shape[0], x.shape[1]
        if position_ids is None:
            position_ids = torch.arange(S, dtype=torch.int32, device=self.device).unsqueeze(0).expand(Bn, S)
        pos = position_ids.to(self.device, torch.int32).contiguous()
        ll = torch.empty(Bn, self.cfg.vocab, dtype=torch.float32, device=self.device)
        sl = (C.c_int * Bn)(*[int(v) for v in seqlens])
        B.check(self.lib.vz_llm_prefill_rows(self.h, int(row0), B.ptr(x), Bn, S, sl, B.ptr(pos), None, B.ptr(ll), self._s()))
        return ll

    def kv_move_rows(self, src: Sequence[int], dst: Sequence[int], lens: Sequence[int]):
        """first lens[i] cache positions of row src[i] -> row dst[i] (all layers), stream-ordered behind the prefill that wrote them."""
        n = len(src)
        assert n == len(dst) == len(lens) and n >= 1
        arr = lambda v: (C.c_int * n)(*[int(x) for x in v])    # noqa: E731
        B.check(self.lib.vz_llm_kv_move_rows(self.h, n, arr(src), arr(dst), arr(lens), self._s()))

    def decode_set_row(self, row: int, token: int, next_pos: int, ctx_len: int):
        """(re)arm one row of the running decode batch; ctx_len 0 parks it."""
        B.check(self.lib.vz_llm_decode_set_row(self.h, int(row), int(token), int(next_pos), int(ctx_len), self._s()))

    def decode_begin(self, first_ids: torch.Tensor, next_pos: Sequence[int], ctx_len: Sequence[int]):
        ids = first_ids.to(self.device, torch.int32).contiguous().view(-1)
        Bn = ids.numel()
        np_ = (C.c_int * Bn)(*[int(v) for v in next_pos])
        cl = (C.c_int * Bn)(*[int(v) for v in ctx_len])
        self._dec_keep = ids
        B.check(self.lib.vz_llm_decode_begin(self.h, Bn, B.ptr(ids), np_, cl, self._s()))
        self._dec_B = Bn

    def decode_steps(self, n: int, out: Optional[torch.Tensor] = None, return_logits: bool = False):
        """enqueue n greedy steps; returns int32 [B,n] (device; no host sync) (+ fp32 logits [n,B,V])."""
        Bn = self._dec_B
        if out is None:
            out = torch.empty(Bn, n, dtype=torch.int32, device=self.device)
        lg = torch.empty(n, Bn, self.cfg.vocab, dtype=torch.float32, device=self.device) if return_logits else None
        B.check(self.lib.vz_llm_decode_steps(self.h, n, B.ptr(out), B.ptr(lg), self._s()))
        return (out, lg) if return_logits else out

    def set_sampling(self, on: bool, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, seed: int = 0, first_counter: int = 1):
        """tail of every decode step: greedy argmax (off) or the device-side sampler (sampling.hip); applies from the next decode_begin."""
        B.check(self.lib.vz_llm_decode_sampling(self.h, int(on), float(temperature), int(top_k or 0), float(1.0 if top_p is None else top_p),
                                                int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_counter)))

    def set_ring(self, ring: Optional[torch.Tensor]):
        """host-visible (pinned) int32 ring the step tails write their tokens to: [slots] for a one-row batch or [rows, slots]
        (slot = draw counter mod slots); None = off.  decode_steps refuses a batch with more rows than the ring has."""
        rows = slots = 0
        if ring is not None:
            assert ring.dtype == torch.int32 and ring.is_pinned() and ring.dim() in (1, 2) and ring.is_contiguous() and ring.shape[-1] >= 2
            rows, slots = (1 if ring.dim() == 1 else int(ring.shape[0])), int(ring.shape[-1])
        self._ring_keep = ring
        B.check(self.lib.vz_llm_decode_ring(self.h, B.ptr(ring), slots, rows))

    def check_async(self):
        """raise if a bounded device-side wait expired since the last check (outputs invalid): the stream-K fix-up of
        the 256^2 GEMM (its tile is NaN, never a stale sum)."""
        err = C.c_int(0)
        B.check(self.lib.vz_engine_async_error(self.h, C.byref(err)))
        if err.value:
            what = {B.VZ_ASYNC_STREAMK: "stream-K fix-up of the 256^2 GEMM (another launch shared its tickets?)",
                    B.VZ_ASYNC_PERSIST: "phase hand-off of the persistent decode-token kernel"}.get(err.value, f"code {err.value}")
            raise RuntimeError(f"vz_hip: a bounded device-side wait expired ({what}); outputs since the last check are invalid")

    def persist_mode(self) -> bool:
        """True if the last decode_steps ran every token as one resident grid (decode_persist.hip)."""
        m = C.c_int(0)
        B.check(self.lib.vz_test_persist_poke(self.h, -1, 0, C.byref(m), self._s()))
        return bool(m.value)

    def persist_poke(self, word: int, value: int):
        """TEST HOOK: preset arrival counter `word` (0..7; 8 = abort) of the persistent decode-token kernel for the next decode_steps."""
        B.check(self.lib.vz_test_persist_poke(self.h, int(word), int(value) & 0xFFFFFFFF, None, self._s()))

    def decode_mode(self):
        """(graph replayed?, RCCL collectives inside the graph?) of the last decode_steps call."""
        g, c = C.c_int(0), C.c_int(0)
        B.check(self.lib.vz_llm_decode_mode(self.h, C.byref(g), C.byref(c)))
        return bool(g.value), bool(c.value)

    # profiling hooks (bench.py roofline leg)
    def prof_enable(self, on: bool, klass: int = -1):
        B.check(self.lib.vz_prof_enable(self.h, int(on), klass))

    def prof_read(self) -> Tuple[int, float]:
        n, ms = C.c_long(), C.c_double()
        B.check(self.lib.vz_prof_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value
