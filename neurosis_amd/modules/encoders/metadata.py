"""`ConcatTimestepEmbedderND` (name and behaviour of reference `modules/encoders/metadata.py:14-36`): SDXL's original-size /
crop / target-size conditioning.  Every scalar of a sample gets its own sinusoidal embedding (the nk_timestep_embedding
kernel via `Timestep`), and a sample's embeddings are laid side by side: (B, n) -> (B, n * outdim)."""
from __future__ import annotations

import torch
from torch import Tensor

from ..diffusion.openaimodel import Timestep
from .embedding import AbstractEmbModel


class ConcatTimestepEmbedderND(AbstractEmbModel):
    def __init__(self, outdim, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.outdim = outdim
        self.timestep = Timestep(outdim)

    def forward(self, x):
        values = torch.stack(x, dim=-1) if isinstance(x, list) else x
        if values.ndim == 1:
            values = values.unsqueeze(1)
        if values.ndim != 2:
            raise ValueError(f"Expected 2D input, got {values.ndim}D")
        batch, count = values.shape
        return self.timestep(values.flatten()).reshape(batch, count * self.outdim)
