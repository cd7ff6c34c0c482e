"""Denoiser / DiscreteDenoiser: mirror of neurosis.modules.diffusion.denoiser (denoiser.py:17-97).

`forward` is the reference's generic path (works with any network module; a handful of [B]- and latent-sized
elementwise ops).  `coefficients` exposes the same per-sample scalars to the fused HIP training path
(StandardDiffusionLoss), where the scaling is folded into the nk_edm_prepare / nk_edm_loss kernels.
"""
from __future__ import annotations

from typing import Union

import torch
from torch import Tensor, nn

from .denoiser_preconditioning import DenoiserPreconditioning
from .discretization import Discretization
from .util import append_dims


class Denoiser(nn.Module):
    def __init__(self, preconditioning: DenoiserPreconditioning):
        super().__init__()
        self.preconditioning = preconditioning

    def possibly_quantize_sigma(self, sigma: Tensor) -> Tensor:
        return sigma

    def possibly_quantize_c_noise(self, c_noise: Tensor) -> Tensor:
        return c_noise

    def coefficients(self, sigma: Tensor):
        """(c_skip, c_out, c_in, c_noise) as [B] tensors, exactly as forward computes them (denoiser.py:37-47)."""
        sigma = self.possibly_quantize_sigma(sigma)
        c_skip, c_out, c_in, c_noise = self.preconditioning(sigma)
        return c_skip, c_out, c_in, self.possibly_quantize_c_noise(c_noise.reshape(sigma.shape))

    def forward(self, network: nn.Module, inputs: Tensor, sigma: Tensor, cond: dict, output_mode: str = "D", **additional_model_inputs) -> Tensor:
        sigma = self.possibly_quantize_sigma(sigma)
        sigma_shape = sigma.shape
        sigma = append_dims(sigma, inputs.ndim)
        c_skip, c_out, c_in, c_noise = self.preconditioning(sigma)
        c_noise = self.possibly_quantize_c_noise(c_noise.reshape(sigma_shape))
        c_in, c_out, c_skip = c_in.to(inputs.dtype), c_out.to(inputs.dtype), c_skip.to(inputs.dtype)
        net_outputs = network(inputs * c_in, c_noise, cond, **additional_model_inputs)
        if output_mode == "F":
            return net_outputs
        return net_outputs * c_out + inputs * c_skip


class DiscreteDenoiser(Denoiser):
    """denoiser.py:60-97: sigma and c_noise snapped to the nearest entry of the discretisation's table."""

    def __init__(self, preconditioning: DenoiserPreconditioning, num_idx: int, discretization: Discretization, do_append_zero: bool = False,
                 quantize_c_noise: bool = True, flip: bool = False):
        super().__init__(preconditioning)
        self.num_idx = num_idx
        self.quantize_c_noise = quantize_c_noise
        self.do_append_zero = do_append_zero
        self.flip = flip
        sigmas = discretization(self.num_idx, do_append_zero=self.do_append_zero, flip=self.flip).detach()
        self.register_buffer("sigmas", sigmas, persistent=False)
        self.register_buffer("log_sigmas", sigmas.log(), persistent=False)

    def sigma_to_idx(self, sigma: Tensor) -> Tensor:
        dists = sigma - self.sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx: Union[Tensor, int]) -> Tensor:
        return self.sigmas[idx]

    def possibly_quantize_sigma(self, sigma: Tensor) -> Tensor:
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise: Tensor) -> Tensor:
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise
