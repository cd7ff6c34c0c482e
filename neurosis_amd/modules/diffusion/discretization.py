"""Sigma tables (host side, fp32).  Class names / call signature of `neurosis.modules.diffusion.discretization`.

Two reference behaviours are kept on purpose: the call's `do_append_zero` ARGUMENT is ignored in favour of
the constructor attribute (Q1: the SDXL table therefore has 1 001 entries), and the legacy DDPM table holds the same
values as the reference's but without its autograd graph (Q5)."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import Tensor

from .util import make_beta_schedule


def generate_roughly_equally_spaced_steps(num_substeps: int, max_step: int) -> np.ndarray:
    """`num_substeps` ascending step indices ending at max_step - 1 (reference discretization.py:13-14)."""
    descending = np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)
    return descending[::-1]


class Discretization:
    def __init__(self, do_append_zero: bool = True):
        self.do_append_zero = do_append_zero

    def get_sigmas(self, n: int, device) -> Tensor:
        raise NotImplementedError(f"{type(self).__name__} does not define a sigma table")

    def __call__(self, n: int, do_append_zero: bool = True, device="cpu", flip: bool = False) -> Tensor:
        table = self.get_sigmas(n, device=device)
        if self.do_append_zero:                      # (the argument of the same name is deliberately unused: Q1)
            table = torch.cat([table, table.new_zeros(1)])
        return table.flip(0) if flip else table


class EDMcDiscretization(Discretization):
    """log-uniform between sigma_min and sigma_max, descending (reference :43-57)."""

    def __init__(self, sigma_min: float = 0.001, sigma_max: float = 1000.0):
        super().__init__()
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        log_sigmas = torch.linspace(math.log(self.sigma_min), math.log(self.sigma_max), n, dtype=torch.float32)
        return log_sigmas.exp().flip(0).to(device)


class LegacyDDPMDiscretization(Discretization):
    """sigma_t = sqrt((1 - abar_t) / abar_t) of the linear-beta DDPM schedule, descending (reference :149-171)."""

    def __init__(self, linear_start: float = 0.00085, linear_end: float = 0.0120, num_timesteps: int = 1000):
        super().__init__()
        self.num_timesteps = num_timesteps
        self.alphas = 1.0 - make_beta_schedule("linear", num_timesteps, linear_start, linear_end)
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0, dtype=torch.float32)

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        if n > self.num_timesteps:
            raise ValueError(f"n ({n}) must be less than or equal to num_timesteps ({self.num_timesteps})")
        abar = self.alphas_cumprod
        if n < self.num_timesteps:
            abar = abar[generate_roughly_equally_spaced_steps(n, self.num_timesteps).copy()]
        table = ((1 - abar) / abar) ** 0.5
        return table.flip(0).to(device, dtype=torch.float32)
