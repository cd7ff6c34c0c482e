"""Scalar helpers of the k-diffusion style samplers (names of `neurosis.modules.diffusion.sampling.utils`, :9-94).

Everything here is [B]-sized or host-side arithmetic; the latent-sized work of a sampling step lives in csrc/sampling.hip.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
from torch import Tensor


def _per_sample(v: Tensor, like: Tensor) -> Tensor:
    """[B] -> [B, 1, 1, ...] matching `like`"""
    return v.reshape(v.shape + (1,) * (like.ndim - v.ndim))


# -- time changes ------------------------------------------------------------------------------------
def to_neg_log_sigma(sigma: Tensor) -> Tensor:
    """t = -log(sigma): the time variable of the DPM-Solver++ family"""
    return -torch.log(sigma)


def to_sigma(neg_log_sigma: Tensor) -> Tensor:
    return torch.exp(-neg_log_sigma)


def to_d(x: Tensor, sigma: Tensor, denoised: Tensor) -> Tensor:
    """dx/dsigma of the probability-flow ODE in the Karras et al. parametrisation"""
    return (x - denoised) / _per_sample(sigma, x)


# -- step sizes ----------------------------------------------------------------------------------------
def get_ancestral_step(sigma_from: Tensor, sigma_to: Tensor, eta: Optional[float] = 1.0):
    """Split the move sigma_from -> sigma_to into a deterministic part down to `sigma_down` and `sigma_up` of fresh noise,
    sigma_down^2 + sigma_up^2 = sigma_to^2.  Returns (sigma_down, sigma_up); eta = 0 is the deterministic sampler."""
    if not eta:
        return sigma_to, 0.0
    var_to, var_from = sigma_to**2, sigma_from**2
    sigma_up = torch.min(sigma_to, eta * (var_to * (var_from - var_to) / var_from) ** 0.5)
    return (var_to - sigma_up**2) ** 0.5, sigma_up


def linear_multistep_coeff(order: int, t, i: int, j: int, epsrel: float = 1e-4) -> float:
    """Adams-Bashforth weight of the j-th most recent derivative for the interval [t_i, t_{i+1}]: the integral of the
    Lagrange basis polynomial through the last `order` nodes (scipy quad, as the reference)."""
    from scipy import integrate

    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")
    node = t[i - j]
    others = [t[i - k] for k in range(order) if k != j]

    def basis(tau: float) -> float:
        value = 1.0
        for other in others:
            value *= (tau - other) / (node - other)
        return value

    return integrate.quad(basis, t[i], t[i + 1], epsrel=epsrel)[0]


def default_noise_sampler(x: Tensor) -> Callable:
    return lambda sigma, sigma_next: torch.randn_like(x)


# -- continuous schedules: n descending levels followed by a final 0 ---------------------------------
def _finish(levels: Tensor, device) -> Tensor:
    return torch.cat((levels, levels.new_zeros(1))).to(device)


def get_sigmas_karras(n: int, sigma_min: float, sigma_max: float, rho: float = 7.0, device="cpu") -> Tensor:
    """Karras et al. (2022) eq. 5: uniform in sigma^(1/rho)"""
    hi, lo = sigma_max ** (1 / rho), sigma_min ** (1 / rho)
    return _finish((hi + torch.linspace(0, 1, n, device=device) * (lo - hi)) ** rho, device)


def get_sigmas_exponential(n: int, sigma_min: float, sigma_max: float, device="cpu") -> Tensor:
    """uniform in log sigma"""
    return _finish(torch.linspace(math.log(sigma_max), math.log(sigma_min), n, device=device).exp(), device)


def get_sigmas_polyexponential(n: int, sigma_min: float, sigma_max: float, rho: float = 1.0, device="cpu") -> Tensor:
    """polynomial (degree rho) in log sigma"""
    span = math.log(sigma_max) - math.log(sigma_min)
    return _finish(torch.exp(torch.linspace(1, 0, n, device=device) ** rho * span + math.log(sigma_min)), device)


def get_sigmas_vp(n: int, beta_d: float = 19.9, beta_min: float = 0.1, eps_s: float = 1e-3, device="cpu") -> Tensor:
    """continuous variance-preserving schedule of Song et al."""
    t = torch.linspace(1, eps_s, n, device=device)
    return _finish(torch.sqrt(torch.exp(beta_d * t**2 / 2 + beta_min * t) - 1), device)
