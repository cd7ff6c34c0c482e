"""Input / output scalings of the denoiser (host side, [B]-sized tensors).

API of `neurosis.modules.diffusion.denoiser_preconditioning` (reference :8-105): an object called with sigma returns
`(c_skip, c_out, c_in, c_noise)`; the `get_c_*` accessors exist because the reference exposes them.  Here every variant
is ONE `scalings(sigma)` method and the accessors are derived from it.

    D(x; sigma) = c_skip * x + c_out * F(c_in * x; c_noise)
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

Scalings = Tuple[Tensor, Tensor, Tensor, Tensor]


class DenoiserPreconditioning:
    def scalings(self, sigma: Tensor) -> Scalings:
        raise NotImplementedError(f"{type(self).__name__} does not define scalings()")

    def __call__(self, sigma: Tensor) -> Scalings:
        return self.scalings(sigma)

    def get_c_skip(self, sigma: Tensor) -> Tensor:
        return self.scalings(sigma)[0]

    def get_c_out(self, sigma: Tensor) -> Tensor:
        return self.scalings(sigma)[1]

    def get_c_in(self, sigma: Tensor) -> Tensor:
        return self.scalings(sigma)[2]

    def get_c_noise(self, sigma: Tensor) -> Tensor:
        return self.scalings(sigma)[3]

    def get_snr(self, sigma: Tensor) -> Tensor:
        return sigma.square().reciprocal()


def _inv_norm(sigma: Tensor, offset: float) -> Tensor:
    """1 / sqrt(sigma^2 + offset)"""
    return (sigma.square() + offset).sqrt().reciprocal()


class EpsPreconditioning(DenoiserPreconditioning):
    """The network predicts the noise: x0 = x - sigma * eps(x / sqrt(sigma^2 + 1); sigma)   (reference :33-44)."""

    def scalings(self, sigma: Tensor) -> Scalings:
        return torch.ones_like(sigma), sigma.neg(), _inv_norm(sigma, 1.0), sigma.clone()


class VPreconditioning(DenoiserPreconditioning):
    """v-prediction (reference :47-52)."""

    def scalings(self, sigma: Tensor) -> Scalings:
        inv = _inv_norm(sigma, 1.0)
        return (sigma.square() + 1.0).reciprocal(), sigma.neg() * inv, inv, sigma.clone()


class EDMPreconditioning(DenoiserPreconditioning):
    """Karras et al. 2022, table 1 (reference :60-77)."""

    def __init__(self, sigma_data: float = 1.0):
        self.sigma_data = sigma_data

    def scalings(self, sigma: Tensor) -> Scalings:
        var = self.sigma_data ** 2
        inv = _inv_norm(sigma, var)
        return var / (sigma.square() + var), sigma * self.sigma_data * inv, inv, sigma.log() * 0.25


class VPreconditioningWithEDMcNoise(VPreconditioning):
    """v-prediction with the EDM noise conditioning c_noise = ln(sigma) / 4 (reference :55-57)."""

    def scalings(self, sigma: Tensor) -> Scalings:
        c_skip, c_out, c_in, _ = super().scalings(sigma)
        return c_skip, c_out, c_in, sigma.log() * 0.25


class RectifiedFlowXLPreconditioning(DenoiserPreconditioning):
    """Rectified flow in the sigma = t / (1 - t) parametrisation (reference :77-90): the input is rescaled to unit variance
    for the interpolant (1 - t) x + t eps, the network is conditioned on 1000 t."""

    def scalings(self, sigma: Tensor) -> Scalings:
        t = sigma / (1.0 + sigma)
        signal = 1.0 / (1.0 + sigma)
        return torch.ones_like(sigma), sigma.neg(), signal / (signal**2.0 + t**2.0) ** 0.5, 1000.0 * t


class RectifiedFlowComfyPreconditioning(DenoiserPreconditioning):
    """Rectified flow with sigma = t itself (reference :93-105)."""

    def scalings(self, sigma: Tensor) -> Scalings:
        return torch.ones_like(sigma), sigma.neg(), (sigma**2.0 + (1.0 - sigma) ** 2.0) ** -0.5, 1000.0 * sigma
