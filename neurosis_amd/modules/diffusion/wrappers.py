"""Mirror of neurosis.modules.diffusion.wrappers (wrappers.py:7-40)."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .openaimodel import UNetModel


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model: UNetModel, compile_model: bool = False, **kwargs):
        super().__init__()
        if compile_model:
            raise NotImplementedError("compile_model: the MI355X path launches HIP kernels / hipGraphs directly; there is no tracing compiler")
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    def forward(self, x: Tensor, t: Tensor, c: dict, **kwargs) -> Tensor:
        concat = c.get("concat", None)
        if concat is not None and concat.numel() > 0:
            x = torch.cat((x, concat.type_as(x)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), **kwargs)
