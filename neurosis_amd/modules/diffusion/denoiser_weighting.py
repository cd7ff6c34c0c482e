"""Per-sample loss weights w(sigma) (host side).  Class names and call signature of
`neurosis.modules.diffusion.denoiser_weighting` (reference :16-101, the variants the diffusion configs use)."""
from __future__ import annotations

import torch
from torch import Tensor


class DenoiserWeighting:
    def __call__(self, sigma: Tensor) -> Tensor:
        raise NotImplementedError(f"{type(self).__name__} does not define a weight")


class UnitWeighting(DenoiserWeighting):
    """w = 1"""

    def __call__(self, sigma: Tensor) -> Tensor:
        return torch.ones_like(sigma)


class EpsWeighting(DenoiserWeighting):
    """w = sigma^-2: an MSE on x0 becomes an MSE on the predicted noise (reference :22-25)."""

    def __call__(self, sigma: Tensor) -> Tensor:
        return sigma.pow(-2.0)


class EDMWeighting(DenoiserWeighting):
    """w = (sigma^2 + sigma_data^2) / (sigma * sigma_data)^2   (Karras et al. 2022; reference :28-35)."""

    def __init__(self, sigma_data: float = 1.0):
        self.sigma_data = sigma_data

    def __call__(self, sigma: Tensor) -> Tensor:
        sd = self.sigma_data
        return (sigma.square() + sd * sd) / (sigma * sd).square()
