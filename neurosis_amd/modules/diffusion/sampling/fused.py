"""The denoiser the engine hands to a sampler on MI355X.

Called like the reference's `denoiser_cb(inputs, sigma, c)` closure (models/diffusion.py:306-310) it is exactly that.  A
sampler from this package additionally finds `guided()` / `euler()`: the same evaluation with the elementwise work around the
network moved into the nk_sample_* kernels (csrc/sampling.hip):

    reference, per step (CFG):  cat([x]*2), cat([s]*2), cat(uc, c) | x*c_in | UNet | F*c_out + x*c_skip | chunk, u + s(c-u) |
                                (x - D)/sigma | x + dt*d                               ~12 launches over the latents, 2B-sized
    here:                       nk_sample_prepare | UNet (bf16 tokens in, bf16 tokens out) | nk_sample_euler_step

The latents stay fp32 NCHW at the API; the network side never leaves channels-last bf16 tokens, so the NCHW<->NHWC
transposes of the generic path disappear as well.
"""
from __future__ import annotations

import torch
from torch import Tensor

from .... import ops
from ....lib import call
from ....nn import as_tokens
from ....ops import BF16, Img
from ...guidance import Guider, IdentityGuider, VanillaCFG
from ..denoiser import Denoiser
from ..wrappers import OpenAIWrapper


class FusedDenoiser:
    def __init__(self, network, denoiser: Denoiser, **model_kwargs):
        self.network, self.denoiser, self.model_kwargs = network, denoiser, model_kwargs

    # the reference's callback ---------------------------------------------------------------------
    def __call__(self, inputs: Tensor, sigma: Tensor, c: dict) -> Tensor:
        return self.denoiser(self.network, inputs, sigma, c, "D", **self.model_kwargs)

    # the fused route ------------------------------------------------------------------------------
    def supports(self, x: Tensor, guider: Guider, cond: dict) -> bool:
        if type(guider) not in (VanillaCFG, IdentityGuider) or not isinstance(self.network, OpenAIWrapper):
            return False
        return x.dim() == 4 and x.dtype == torch.float32 and self.network.fused_unet(x, cond, self.model_kwargs) is not None

    def _network(self, x: Tensor, sigma: Tensor, cond: dict, uc: dict, guider: Guider):
        """one UNet evaluation on rep stacked copies of c_in * x -> (bf16 output tokens Img, c_skip[B], c_out[B])"""
        B, C, H, W = x.shape
        rep = guider.rep
        unet = self.network.diffusion_model
        c_skip, c_out, c_in, c_noise = (t.contiguous() for t in self.denoiser.coefficients(sigma))
        c_skip, c_out, c_in = c_skip.float(), c_out.float(), c_in.float()
        cpad = (C + 7) // 8 * 8
        net_in = torch.empty(rep * B * H * W, cpad, dtype=BF16, device=x.device)
        call("nk_sample_prepare", x.data_ptr(), c_in.data_ptr(), net_in.data_ptr(), B, C, H * W, cpad, rep, ops._stream())

        def stacked(key):
            value = cond.get(key)
            if value is None:
                return None
            return as_tokens(value if rep == 1 else torch.cat((uc[key], value), 0))

        timesteps = c_noise if rep == 1 else torch.cat((c_noise, c_noise))
        out, _ = unet.fwd(Img(net_in, rep * B, H, W), timesteps, stacked("crossattn"), stacked("vector"))
        return out, c_skip, c_out

    def guided(self, x: Tensor, sigma: Tensor, cond: dict, uc: dict, guider: Guider) -> Tensor:
        """guider(denoiser(*guider.prepare_inputs(x, sigma, cond, uc)), sigma) as fp32 NCHW"""
        x = x.contiguous()
        B, C, H, W = x.shape
        out, c_skip, c_out = self._network(x, sigma, cond, uc, guider)
        denoised = torch.empty_like(x)
        call("nk_sample_denoise", out.t.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), float(getattr(guider, "scale", 1.0)),
             denoised.data_ptr(), B, C, H * W, out.C, guider.rep, ops._stream())
        return denoised

    def euler(self, x: Tensor, sigma_hat: Tensor, next_sigma: Tensor, cond: dict, uc: dict, guider: Guider) -> Tensor:
        """x + (next_sigma - sigma_hat) * (x - D) / sigma_hat with D as in `guided`"""
        x = x.contiguous()
        B, C, H, W = x.shape
        out, c_skip, c_out = self._network(x, sigma_hat, cond, uc, guider)
        x_next = torch.empty_like(x)
        sh, sn = sigma_hat.float().contiguous(), next_sigma.float().contiguous()
        call("nk_sample_euler_step", out.t.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), sh.data_ptr(), sn.data_ptr(),
             float(getattr(guider, "scale", 1.0)), x_next.data_ptr(), None, B, C, H * W, out.C, guider.rep, ops._stream())
        return x_next
