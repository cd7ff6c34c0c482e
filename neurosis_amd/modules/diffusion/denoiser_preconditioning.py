"""c_skip / c_out / c_in / c_noise (host-side, [B] scalars): mirror of neurosis.modules.diffusion.denoiser_preconditioning."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Tuple

import torch
from torch import Tensor


class DenoiserPreconditioning(ABC):
    """denoiser_preconditioning.py:8-31."""

    def __call__(self, sigma: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
        return self.get_c_skip(sigma), self.get_c_out(sigma), self.get_c_in(sigma), self.get_c_noise(sigma)

    @abstractmethod
    def get_c_skip(self, sigma: Tensor) -> Tensor: ...

    @abstractmethod
    def get_c_out(self, sigma: Tensor) -> Tensor: ...

    @abstractmethod
    def get_c_in(self, sigma: Tensor) -> Tensor: ...

    @abstractmethod
    def get_c_noise(self, sigma: Tensor) -> Tensor: ...

    def get_snr(self, sigma: Tensor) -> Tensor:
        return 1 / sigma**2.0


class EpsPreconditioning(DenoiserPreconditioning):
    """denoiser_preconditioning.py:33-44."""

    def get_c_skip(self, sigma: Tensor) -> Tensor:
        return torch.ones_like(sigma, device=sigma.device)

    def get_c_out(self, sigma: Tensor) -> Tensor:
        return -sigma

    def get_c_in(self, sigma: Tensor) -> Tensor:
        return 1.0 / (sigma**2.0 + 1.0) ** 0.5

    def get_c_noise(self, sigma: Tensor) -> Tensor:
        return sigma.clone()


class VPreconditioning(EpsPreconditioning):
    """denoiser_preconditioning.py:47-52."""

    def get_c_skip(self, sigma: Tensor) -> Tensor:
        return 1.0 / (sigma**2 + 1.0)

    def get_c_out(self, sigma: Tensor) -> Tensor:
        return -sigma / (sigma**2 + 1.0) ** 0.5


class EDMPreconditioning(DenoiserPreconditioning):
    """denoiser_preconditioning.py:60-77."""

    def __init__(self, sigma_data: float = 0.5):
        self.sigma_data = sigma_data

    def get_c_skip(self, sigma: Tensor) -> Tensor:
        return self.sigma_data**2 / (sigma**2 + self.sigma_data**2)

    def get_c_out(self, sigma: Tensor) -> Tensor:
        return sigma * self.sigma_data / (sigma**2 + self.sigma_data**2) ** 0.5

    def get_c_in(self, sigma: Tensor) -> Tensor:
        return 1 / (sigma**2 + self.sigma_data**2) ** 0.5

    def get_c_noise(self, sigma: Tensor) -> Tensor:
        return 0.25 * sigma.log()
