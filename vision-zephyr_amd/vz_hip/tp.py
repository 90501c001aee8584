"""Tensor-parallel sharding plan for Zephyr-7B over the GPUs of one node.

Host logic only (which slice of which reference tensor a rank owns); the collectives are RCCL
all-reduce / all-gather over xGMI.  32 query heads / 8 KV heads / 14336 MLP columns divide exactly
by 2, 4 and 8:

  column-parallel (output features split, no communication):   q_proj, k_proj, v_proj, gate_proj, up_proj
  row-parallel    (input features split, partial sums all-reduced): o_proj, down_proj
  vocab-parallel  (rows of lm_head split, logits all-gathered / arg-max reduced): lm_head
  replicated: embed_tokens, every norm weight, CLIP, Q-Former (those are tile-data-parallel instead)

=> 2 all-reduces of [B,S,4096] per layer, KV cache sharded by KV head.
"""
from __future__ import annotations

import re
from typing import Tuple

import torch

COL, ROW, VOCAB, REPL = "col", "row", "vocab", "replicate"


def kind_of(name: str) -> str:
    if re.search(r"self_attn\.(q|k|v)_proj\.weight$", name) or re.search(r"mlp\.(gate|up)_proj\.weight$", name):
        return COL
    if re.search(r"self_attn\.o_proj\.weight$", name) or re.search(r"mlp\.down_proj\.weight$", name):
        return ROW
    if name == "lm_head.weight":
        return VOCAB
    return REPL


def check_divisible(cfg, world: int):
    if cfg.n_kv_heads % world or cfg.n_heads % world or cfg.inter % world:
        raise ValueError(f"tensor-parallel degree {world} must divide kv heads ({cfg.n_kv_heads}), heads ({cfg.n_heads}) "
                         f"and the MLP width ({cfg.inter})")


def shard(name: str, t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's slice of a reference-named tensor.  q/k/v and gate/up are split by whole heads / contiguous
    column blocks, so concatenating the ranks' outputs in rank order reproduces the unsharded layout."""
    k = kind_of(name)
    if k == REPL or world == 1:
        return t
    if k == COL:
        n = t.shape[0] // world
        return t[rank * n:(rank + 1) * n]
    if k == ROW:
        n = t.shape[1] // world
        return t[:, rank * n:(rank + 1) * n]
    # vocab-parallel: ceil-split so that a 32001-row table (tokenizer + <im_patch>) still shards
    n = (t.shape[0] + world - 1) // world
    return t[rank * n:min(t.shape[0], (rank + 1) * n)]


def vocab_range(vocab: int, rank: int, world: int) -> Tuple[int, int]:
    n = (vocab + world - 1) // world
    return rank * n, min(vocab, (rank + 1) * n)


def allreduce_bytes_per_layer(cfg, tokens: int) -> int:
    """bf16 payload of the two all-reduces of one decoder layer (16.8 MB at S=2048)."""
    return 2 * tokens * cfg.hidden * 2


# ---- tile data parallelism for CLIP + fusion + Q-Former (first row) ----
def local_tiles(n_tiles: int, rank: int, world: int):
    """tiles of a batch dealt round-robin: tile t lives on rank t mod world (5 tiles on 8 GPUs -> 5 GPUs busy)."""
    return list(range(rank, n_tiles, world))


def tiles_per_rank(n_tiles: int, world: int) -> int:
    """slots per rank in the all-gather buffer (the ranks with fewer tiles zero-pad)."""
    return (n_tiles + world - 1) // world


def gathered_index(n_tiles: int, world: int) -> torch.Tensor:
    """row of the flattened all-gather result [world * tiles_per_rank] that holds tile t: rank (t mod world), slot (t div world)."""
    per = tiles_per_rank(n_tiles, world)
    t = torch.arange(n_tiles)
    return (t % world) * per + t // world
