"""Loss weights w(sigma) (host side): mirror of neurosis.modules.diffusion.denoiser_weighting."""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch
from torch import Tensor


class DenoiserWeighting(ABC):
    @abstractmethod
    def __call__(self, sigma: Tensor) -> Tensor: ...


class UnitWeighting(DenoiserWeighting):
    def __call__(self, sigma: Tensor) -> Tensor:
        return torch.ones_like(sigma, device=sigma.device)


class EpsWeighting(DenoiserWeighting):
    """denoiser_weighting.py:22-25."""

    def __call__(self, sigma: Tensor) -> Tensor:
        return sigma**-2.0


class EDMWeighting(DenoiserWeighting):
    """denoiser_weighting.py:28-35."""

    def __init__(self, sigma_data: float = 1.0):
        self.sigma_data = sigma_data

    def __call__(self, sigma: Tensor) -> Tensor:
        return (sigma**2 + self.sigma_data**2) / (sigma * self.sigma_data) ** 2
