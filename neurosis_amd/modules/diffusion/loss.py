"""StandardDiffusionLoss: mirror of neurosis.modules.diffusion.loss (loss.py:20-157).

When the network is an OpenAIWrapper around this package's UNetModel, the l2 loss runs fused for both objectives:
noising + input scaling (nk_edm_prepare), the UNet as one explicit forward/backward chain of HIP kernels,
and output scaling + per-sample weighted MSE + its gradient (nk_edm_loss) -- one autograd node for the whole
loss.  "edm": z_t = x + sigma eps, target x, D = c_skip z_t + c_out F.  "rf" (rectified flow): z_t = (1 - sigma) x + sigma eps,
target eps, the raw network output F (the same two kernels: the prepare kernel is handed (1 - sigma) x, the loss kernel
c_out = 1, c_skip = 0 and the noise as target).  Any other network, and the l1 loss, take the reference's generic route
through Denoiser.forward.
"""
from __future__ import annotations

import random
from typing import Optional

import torch
from torch import Tensor, nn

from ... import ops
from ...lib import call
from ...nn import NkFunction, as_tokens
from ...ops import BF16, Img
from .denoiser import Denoiser
from .denoiser_weighting import DenoiserWeighting
from .openaimodel import UNetModel
from .util import append_dims
from .wrappers import OpenAIWrapper


class DiffusionLoss(nn.Module):
    """loss.py:20-58."""

    def __init__(self, noise_offset: float = 0.0, noise_offset_chance: float = 0.0, *args, **kwargs):
        super().__init__()
        self.noise_offset = min(max(noise_offset, 0.0), 1.0)
        self.noise_offset_chance = min(max(noise_offset_chance, 0.0), 1.0)

    def apply_noise_offset(self, noise: Tensor, inputs: Tensor) -> Tensor:
        if self.noise_offset <= 0:
            return noise
        if self.noise_offset_chance == 1.0 or random.random() < self.noise_offset_chance:
            offset = torch.randn(inputs.shape[:2] + (1,) * (inputs.ndim - 2)).to(noise)
            return noise + self.noise_offset * offset
        return noise

    def forward(self, network, denoiser, conditioner, inputs, batch, return_dict: bool = False):
        cond = conditioner(batch)
        return self._forward(network, denoiser, cond, inputs, batch, return_dict)


class StandardDiffusionLoss(DiffusionLoss):
    def __init__(self, sigma_generator, loss_weighting: DenoiserWeighting, loss_type: str = "l2", snr_gamma: float = 0.0, noise_offset: float = 0.0,
                 noise_offset_chance: float = 0.0, input_keys=[], objective_type: str = "edm"):
        super().__init__(noise_offset, noise_offset_chance)
        self.sigma_generator = sigma_generator
        self.loss_weighting = loss_weighting
        self.snr_gamma = snr_gamma
        self.objective_type = str(objective_type).lower()
        lt = {"mse": "l2", "mae": "l1"}.get(str(loss_type).lower(), str(loss_type).lower())
        if lt not in ("l1", "l2"):
            raise ValueError(f"Unknown loss type {loss_type}")
        self.loss_type = lt
        if self.objective_type not in ("edm", "rf"):
            raise ValueError(f"Unknown objective type: '{objective_type}'")
        if not isinstance(input_keys, list):
            input_keys = [input_keys]
        self.input_keys = set(input_keys)

    # -- fused HIP route ---------------------------------------------------------------------------
    @staticmethod
    def fused_edm(unet: UNetModel, denoiser: Denoiser, weighting, inputs: Tensor, sigmas: Tensor, noise: Tensor, cond: dict, objective: str = "edm"):
        """loss[B] (fp32) for the edm or rf objective with the l2 loss; differentiable w.r.t. the UNet parameters.
        inputs / noise: fp32 NCHW latents; sigmas: [B] fp32 on the same device."""
        B, Cc, H, W = inputs.shape
        dev = inputs.device
        c_skip, c_out, c_in, c_noise = denoiser.coefficients(sigmas)
        w = weighting(sigmas).float().contiguous()
        c_skip, c_out, c_in = (t.float().contiguous() for t in (c_skip, c_out, c_in))
        sig = sigmas.float().contiguous()
        x = inputs.float().contiguous()
        eps = noise.float().contiguous()
        target = x
        if objective == "rf":
            # z_t = (1 - sigma) x + sigma eps; the network's raw output is compared with the noise
            x = (x * (1.0 - sig).reshape(B, 1, 1, 1)).contiguous()
            target = eps
            c_out, c_skip = torch.ones_like(c_out), torch.zeros_like(c_skip)
        context, y = cond.get("crossattn", None), cond.get("vector", None)
        cpad = (Cc + 7) // 8 * 8
        params = [p for p in unet.parameters() if p.requires_grad]

        def run():
            zt = torch.empty_like(x)
            net_in = torch.empty(B * H * W, cpad, dtype=BF16, device=dev)
            call("nk_edm_prepare", x.data_ptr(), eps.data_ptr(), sig.data_ptr(), c_in.data_ptr(), zt.data_ptr(), net_in.data_ptr(), B, Cc, H * W, cpad, ops._stream())
            out, unet_bwd = unet.fwd_graphed(Img(net_in, B, H, W), c_noise, None if context is None else as_tokens(context), None if y is None else as_tokens(y))
            loss = torch.empty(B, dtype=torch.float32, device=dev)
            call("nk_edm_loss", out.t.data_ptr(), zt.data_ptr(), target.data_ptr(), c_out.data_ptr(), c_skip.data_ptr(), w.data_ptr(), loss.data_ptr(), None,
                 B, Cc, H * W, out.C, 1.0, ops._stream())

            def bwd(dloss: Tensor):
                # d loss[b] / d net_out scaled by the upstream gradient of each sample: reuse the loss kernel with w*g
                wg = (w * dloss.float()).contiguous()
                dnet = torch.empty_like(out.t)
                scratch = torch.empty(B, dtype=torch.float32, device=dev)
                call("nk_edm_loss", out.t.data_ptr(), zt.data_ptr(), target.data_ptr(), c_out.data_ptr(), c_skip.data_ptr(), wg.data_ptr(), scratch.data_ptr(),
                     dnet.data_ptr(), B, Cc, H * W, out.C, 1.0, ops._stream())
                unet_bwd(dnet)
                return ()

            return loss, bwd

        if not torch.is_grad_enabled() or not params:
            return run()[0]
        return NkFunction.apply(run, 0, *params)

    def _forward(self, network: nn.Module, denoiser: Denoiser, cond: dict, inputs: Tensor, batch: dict, return_dict: bool = False,
                 sigmas: Optional[Tensor] = None, noise: Optional[Tensor] = None):
        """loss.py:105-151.  `sigmas` / `noise` may be injected (parity tests, benchmarks: SURVEY quirk Q3)."""
        extra_inputs = {k: batch[k] for k in batch if k in self.input_keys}
        t = torch.rand((inputs.shape[0],), dtype=torch.float64)
        if sigmas is None:
            sigmas = self.sigma_generator(inputs.shape[0], t)
        sigmas = sigmas.to(inputs)
        if noise is None:
            noise = torch.randn_like(inputs)
        noise = self.apply_noise_offset(noise, inputs)
        unet = network.fused_unet(inputs, cond, extra_inputs) if isinstance(network, OpenAIWrapper) and self.loss_type == "l2" else None
        if unet is not None:
            loss = self.fused_edm(unet, denoiser, self.loss_weighting, inputs, sigmas, noise, cond, self.objective_type)
        else:
            sigmas_bc = append_dims(sigmas, inputs.ndim)
            if self.objective_type == "rf":
                z_t = (1.0 - sigmas_bc) * inputs + sigmas_bc * noise
                loss = self.get_loss(denoiser(network, z_t, sigmas, cond, "F", **extra_inputs), noise, self.loss_weighting(sigmas))
            else:
                z_t = inputs + sigmas_bc * noise
                loss = self.get_loss(denoiser(network, z_t, sigmas, cond, "D", **extra_inputs), inputs, self.loss_weighting(sigmas))
        if return_dict:
            return loss, {"sigmas": sigmas, "t": t}
        return loss

    def get_loss(self, outputs: Tensor, target: Tensor, weight: Tensor) -> Tensor:
        """loss.py:153-157 with BatchMSELoss / BatchL1Loss (losses/functions.py:65-94): per-sample mean, then the weight."""
        diff = outputs.float() - target.float()
        per_element = diff * diff if self.loss_type == "l2" else diff.abs()
        return per_element.flatten(1).mean(1) * weight.float()
