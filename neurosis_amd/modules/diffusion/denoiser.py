"""Denoiser / DiscreteDenoiser with the reference's class surface (`neurosis.modules.diffusion.denoiser`, :17-97).

Two consumers: the generic `forward` (any network module; a handful of [B]- and latent-sized elementwise ops) and the
fused HIP training path (`StandardDiffusionLoss`), which asks `coefficients()` for the same per-sample scalars and folds
them into the nk_edm_prepare / nk_edm_loss kernels.
"""
from __future__ import annotations

from typing import Union

import torch
from torch import Tensor, nn

from .denoiser_preconditioning import DenoiserPreconditioning
from .discretization import Discretization


class Denoiser(nn.Module):
    def __init__(self, preconditioning: DenoiserPreconditioning):
        super().__init__()
        self.preconditioning = preconditioning

    # hooks the discrete variant overrides -------------------------------------------------------
    def possibly_quantize_sigma(self, sigma: Tensor) -> Tensor:
        return sigma

    def possibly_quantize_c_noise(self, c_noise: Tensor) -> Tensor:
        return c_noise

    def coefficients(self, sigma: Tensor):
        """(c_skip, c_out, c_in, c_noise), each shaped like `sigma`, c_noise already quantised."""
        snapped = self.possibly_quantize_sigma(sigma)
        skip, out, cin, noise = self.preconditioning(snapped)
        return skip, out, cin, self.possibly_quantize_c_noise(noise.reshape(snapped.shape))

    def forward(self, network: nn.Module, inputs: Tensor, sigma: Tensor, cond: dict, output_mode: str = "D", **additional_model_inputs) -> Tensor:
        if output_mode not in ("D", "F"):
            raise ValueError(f"output_mode must be 'D' (denoised) or 'F' (raw network output), got {output_mode!r}")
        if sigma.ndim != 1 or sigma.shape[0] != inputs.shape[0]:
            raise ValueError(f"sigma must hold one value per sample: got {tuple(sigma.shape)} for a batch of {inputs.shape[0]}")
        skip, out, cin, noise = self.coefficients(sigma)
        bshape = (-1,) + (1,) * (inputs.ndim - 1)          # broadcast the [B] scalars over the latent dims
        prediction = network(inputs * cin.reshape(bshape).to(inputs.dtype), noise, cond, **additional_model_inputs)
        if output_mode == "F":                              # raw network output
            return prediction
        return prediction * out.reshape(bshape).to(inputs.dtype) + inputs * skip.reshape(bshape).to(inputs.dtype)


class DiscreteDenoiser(Denoiser):
    """sigma and c_noise are snapped to the nearest entry of the discretisation's table (reference :60-97)."""

    def __init__(self, preconditioning: DenoiserPreconditioning, num_idx: int, discretization: Discretization, do_append_zero: bool = False,
                 quantize_c_noise: bool = True, flip: bool = False):
        super().__init__(preconditioning)
        self.num_idx, self.quantize_c_noise, self.do_append_zero, self.flip = num_idx, quantize_c_noise, do_append_zero, flip
        table = discretization(num_idx, do_append_zero=do_append_zero, flip=flip).detach()
        self.register_buffer("sigmas", table, persistent=False)
        self.register_buffer("log_sigmas", table.log(), persistent=False)

    def sigma_to_idx(self, sigma: Tensor) -> Tensor:
        """index of the nearest table entry (first one on ties), same shape as `sigma`"""
        distance = (self.sigmas.unsqueeze(1) - sigma.reshape(1, -1)).abs()
        return distance.argmin(dim=0).reshape(sigma.shape)

    def extra_repr(self) -> str:
        lo, hi = float(self.sigmas.min()), float(self.sigmas.max())
        return f"num_idx={self.num_idx}, sigma=[{lo:.4g}, {hi:.4g}], quantize_c_noise={self.quantize_c_noise}, flip={self.flip}"

    def idx_to_sigma(self, idx: Union[Tensor, int]) -> Tensor:
        return self.sigmas[idx]

    def possibly_quantize_sigma(self, sigma: Tensor) -> Tensor:
        return self.sigmas[self.sigma_to_idx(sigma)]

    def possibly_quantize_c_noise(self, c_noise: Tensor) -> Tensor:
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise
