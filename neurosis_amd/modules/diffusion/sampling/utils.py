"""Scalar helpers of the k-diffusion style samplers (`neurosis.modules.diffusion.sampling.utils`, :9-94)."""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor


def _per_sample(v: Tensor, like: Tensor) -> Tensor:
    """[B] -> [B, 1, 1, ...] matching `like`"""
    return v.reshape(v.shape + (1,) * (like.ndim - v.ndim))


def _with_final_zero(sigmas: Tensor) -> Tensor:
    return torch.cat((sigmas, sigmas.new_zeros(1)))


def default_noise_sampler(x: Tensor):
    return lambda sigma, sigma_next: torch.randn_like(x)


def linear_multistep_coeff(order: int, t, i: int, j: int, epsrel: float = 1e-4) -> float:
    """integral over [t_i, t_{i+1}] of the j-th Lagrange basis polynomial through the last `order` nodes"""
    from scipy import integrate

    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")
    others = [k for k in range(order) if k != j]

    def basis(tau: float) -> float:
        value = 1.0
        for k in others:
            value *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return value

    return integrate.quad(basis, t[i], t[i + 1], epsrel=epsrel)[0]


def get_ancestral_step(sigma_from: Tensor, sigma_to: Tensor, eta: Optional[float] = 1.0):
    """(sigma_down, sigma_up): the level to step down to and the amount of fresh noise to add afterwards"""
    if not eta:
        return sigma_to, 0.0
    var_to, var_from = sigma_to**2, sigma_from**2
    sigma_up = torch.min(sigma_to, eta * (var_to * (var_from - var_to) / var_from) ** 0.5)
    return (var_to - sigma_up**2) ** 0.5, sigma_up


def to_d(x: Tensor, sigma: Tensor, denoised: Tensor) -> Tensor:
    """Karras ODE derivative dx/dsigma"""
    return (x - denoised) / _per_sample(sigma, x)


def to_neg_log_sigma(sigma: Tensor) -> Tensor:
    return -torch.log(sigma)


def to_sigma(neg_log_sigma: Tensor) -> Tensor:
    return torch.exp(-neg_log_sigma)


# -- continuous schedules (all return n + 1 values, the last one 0) ---------------------------------
def get_sigmas_vp(n: int, beta_d: float = 19.9, beta_min: float = 0.1, eps_s: float = 1e-3, device="cpu") -> Tensor:
    t = torch.linspace(1, eps_s, n, device=device)
    return _with_final_zero(torch.sqrt(torch.exp(beta_d * t**2 / 2 + beta_min * t) - 1))


def get_sigmas_karras(n: int, sigma_min: float, sigma_max: float, rho: float = 7.0, device="cpu") -> Tensor:
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    ramp = torch.linspace(0, 1, n, device=device)
    return _with_final_zero((hi + ramp * (lo - hi)) ** rho).to(device)


def get_sigmas_exponential(n: int, sigma_min: float, sigma_max: float, device="cpu") -> Tensor:
    return _with_final_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp())


def get_sigmas_polyexponential(n: int, sigma_min: float, sigma_max: float, rho: float = 1.0, device="cpu") -> Tensor:
    ramp = torch.linspace(1, 0, n, device=device) ** rho
    return _with_final_zero(torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min)))
