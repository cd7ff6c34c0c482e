"""Network wrappers with the reference's names (`neurosis.modules.diffusion.wrappers`, :7-40): the denoiser calls
`wrapper(x, t, cond)`, the wrapper unpacks the conditioning dict into the UNet's keyword arguments."""
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
    """cond keys: "concat" (extra input channels), "crossattn" (context tokens), "vector" (pooled / size embedding)."""

    def fused_unet(self, inputs: Tensor, cond: dict, extra_inputs: dict):
        """The bare UNetModel when this call can take the fused HIP training path (device latents, no channel-concat
        conditioning, no extra network inputs: nk_edm_prepare feeds the UNet's first conv directly), else None."""
        unet = self.diffusion_model
        plain = not extra_inputs and cond.get("concat") is None
        return unet if isinstance(unet, UNetModel) and inputs.is_cuda and plain else None

    def forward(self, x: Tensor, t: Tensor, c: dict, **kwargs) -> Tensor:
        # other keys in `c` are ignored, as in the reference
        extra = c.get("concat")
        net_in = x if extra is None or extra.numel() == 0 else torch.cat([x, extra.to(x.dtype)], dim=1)
        return self.diffusion_model(net_in, timesteps=t, context=c.get("crossattn"), y=c.get("vector"), **kwargs)
