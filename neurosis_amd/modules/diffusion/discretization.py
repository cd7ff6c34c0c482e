"""Sigma tables (host side, fp32/fp64 scalars): mirror of neurosis.modules.diffusion.discretization."""
from __future__ import annotations

from abc import ABC, abstractmethod
from math import log

import numpy as np
import torch
from torch import Tensor

from .util import append_zero, make_beta_schedule


def generate_roughly_equally_spaced_steps(num_substeps: int, max_step: int) -> np.ndarray:
    return np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)[::-1]


class Discretization(ABC):
    """discretization.py:17-40.  NOTE: __call__ ignores its do_append_zero argument and uses the
    instance attribute (default True); reproduced so the sigma table has the reference's 1001 entries."""

    def __init__(self, do_append_zero: bool = True):
        super().__init__()
        self.do_append_zero = do_append_zero

    def __call__(self, n: int, do_append_zero: bool = True, device="cpu", flip: bool = False) -> Tensor:
        sigmas = self.get_sigmas(n, device=device)
        if self.do_append_zero:
            sigmas = append_zero(sigmas)
        if flip:
            sigmas = sigmas.flip((0,))
        return sigmas

    @abstractmethod
    def get_sigmas(self, n: int, device) -> Tensor:
        raise NotImplementedError("Abstract base class was called ;_;")


class EDMcDiscretization(Discretization):
    """discretization.py:43-57."""

    def __init__(self, sigma_min: float = 0.001, sigma_max: float = 1000.0):
        super().__init__()
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        sigmas = torch.linspace(log(self.sigma_min), log(self.sigma_max), n, dtype=torch.float32).exp()
        return sigmas.flip(0).to(device)


class LegacyDDPMDiscretization(Discretization):
    """discretization.py:149-171.  The table is detached (the reference's carries an autograd graph that breaks a
    second backward; the values are identical)."""

    def __init__(self, linear_start: float = 0.00085, linear_end: float = 0.0120, num_timesteps: int = 1000):
        super().__init__()
        self.num_timesteps = num_timesteps
        self.alphas = 1.0 - make_beta_schedule("linear", num_timesteps, linear_start, linear_end)
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0, dtype=torch.float32)

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        if n < self.num_timesteps:
            timesteps = generate_roughly_equally_spaced_steps(n, self.num_timesteps)
            alphas_cumprod = self.alphas_cumprod[timesteps.copy()].clone()
        elif n == self.num_timesteps:
            alphas_cumprod = self.alphas_cumprod.clone()
        else:
            raise ValueError(f"n ({n}) must be less than or equal to num_timesteps ({self.num_timesteps})")
        sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
        return sigmas.flip(0).to(device, dtype=torch.float32)
