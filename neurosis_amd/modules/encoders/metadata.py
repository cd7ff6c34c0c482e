"""`ConcatTimestepEmbedderND` (reference `modules/encoders/metadata.py:14-36`): the SDXL size / crop / target-size
conditioning -- every scalar gets its own sinusoidal embedding (`Timestep(outdim)` -> nk_timestep_embedding) and the
embeddings of one sample are concatenated: (B, dims) -> (B, dims * outdim)."""
from __future__ import annotations

import torch
from torch import Tensor

from ..diffusion.openaimodel import Timestep
from .embedding import AbstractEmbModel


class ConcatTimestepEmbedderND(AbstractEmbModel):
    def __init__(self, outdim, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.timestep = Timestep(outdim)
        self.outdim = outdim

    def forward(self, x):
        if isinstance(x, list):
            x = torch.stack(x, dim=-1)
        if x.ndim == 1:
            x = x[:, None]
        if x.ndim != 2:
            raise ValueError(f"Expected 2D input, got {x.ndim}D")
        b, dims = x.shape[0], x.shape[1]
        emb = self.timestep(x.reshape(b * dims))
        return emb.reshape(b, dims * self.outdim)
