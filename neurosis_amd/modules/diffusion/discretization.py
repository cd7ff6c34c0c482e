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


class EDMcSimpleDiscretization(Discretization):
    """n levels picked at equal strides from the top of a fixed num_sigmas-entry log-uniform table, then a literal 0.0
    (reference :60-83; the base class appends ANOTHER zero, as there)."""

    def __init__(self, sigma_min: float = 0.001, sigma_max: float = 1000.0, num_sigmas: int = 1000):
        super().__init__()
        self.sigma_min, self.sigma_max, self.num_sigmas = sigma_min, sigma_max, num_sigmas

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        table = torch.linspace(math.log(self.sigma_min), math.log(self.sigma_max), self.num_sigmas, dtype=torch.float32).exp()
        stride = len(table) / n
        picked = [float(table[-(1 + int(i * stride))]) for i in range(n)]
        return torch.tensor(picked + [0.0]).to(device)


class RectifiedFlowDiscretization(Discretization):
    """t uniform in [start_shift, 1 - end_shift], sigma = t / (1 - t), descending (reference :86-95)"""

    as_ratio = True

    def __init__(self, start_shift: float = 0.0, end_shift: float = 0.001, do_append_zero: bool = False):
        super().__init__(do_append_zero=do_append_zero)
        self.start_shift, self.end_shift = start_shift, end_shift

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        t = torch.linspace(self.start_shift, 1 - self.end_shift, n, dtype=torch.float64)
        return (t / (1.0 - t) if self.as_ratio else t).flip(0).to(device, dtype=torch.float32)


class RectifiedFlowComfyDiscretization(RectifiedFlowDiscretization):
    """sigma = t (reference :98-106)"""

    as_ratio = False


class TanZeroSNRDiscretization(Discretization):
    """sigma = scale * tan(angle), angle uniform in [start_shift, pi/2 - end_shift] (fp64), descending (reference :109-124)"""

    def __init__(self, start_shift: float = 0.001, end_shift: float = 0.001, scale: float = 1.0):
        super().__init__()
        self.start_shift, self.end_shift, self.scale = start_shift, end_shift, scale

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        half_pi = torch.acos(torch.zeros(1, dtype=torch.float64))[0]
        angles = torch.linspace(self.start_shift, half_pi - self.end_shift, n, dtype=torch.float64)
        return torch.tan(angles).mul(self.scale).flip(0).to(device, dtype=torch.float32)


class EDMDiscretization(Discretization):
    """Karras et al. (2022) eq. 5, sigma_max first (reference :127-146)"""

    def __init__(self, sigma_min: float = 0.002, sigma_max: float = 80.0, rho: float = 7.0):
        super().__init__()
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n: int, device="cpu") -> Tensor:
        hi, lo = self.sigma_max ** (1 / self.rho), self.sigma_min ** (1 / self.rho)
        return (hi + torch.linspace(0, 1, n, device=device, dtype=torch.float32) * (lo - hi)) ** self.rho
