"""Q-Former projector facade (ref:vis_zephyr/model/multimodal_projector/builder.py:49-101).

`mm_projector(features, text_embeddings=...)` keeps the reference's call shape; the 8 blocks run in
`vz_qformer`.  Text conditioning arrives per tile, as the reference passes it; rows that repeat the
same text (the tiles of one sample, ref:vis_zephyr/model/vis_zephyr_arch.py:174) are detected by the
caller, which can pass `tile_sample` to share block 0's self-attention across them."""
from __future__ import annotations

import torch


class QFormer:
    num_queries = 32

    def __init__(self, config, owner=None):
        self.hidden_size = config.hidden_size
        self._owner = owner

    @torch.no_grad()
    def forward(self, features, text_embeddings=None, tile_sample=None):
        eng = self._owner.engine
        T = features.shape[0]
        if text_embeddings is None:
            return eng.qformer(features, None, [0] * T)
        if tile_sample is None:     # no sharing information: every tile is its own sample
            return eng.qformer(features, text_embeddings, list(range(T)))
        return eng.qformer(features, text_embeddings, tile_sample)

    __call__ = forward


def build_multimodal_projector(config, **kwargs):
    """always a Q-Former: `mm_projector_type` is ignored by the reference too (Appendix A Q2)."""
    return QFormer(config, **kwargs)
