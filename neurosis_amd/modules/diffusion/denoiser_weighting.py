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


def _logit_normal_density(t: Tensor, logit: Tensor, m: float, s: float) -> Tensor:
    """density of t when logit(t) ~ N(m, s^2), up to the reference's constant: 1 / (s sqrt(2 pi)) / (t (1 - t)) * exp(...)"""
    two_pi = 4.0 * torch.acos(torch.zeros(1, dtype=torch.float64))[0]
    return (1 / (s * two_pi**0.5)) * (1 / (t * (1.0 - t))) * torch.exp(-0.5 * (logit - m) ** 2 / s**2)


class RectifiedFlowWeighting(DenoiserWeighting):
    """conditional-flow-matching weight 1 / (1 - t)^2 times the logit-normal density of t = sigma / (1 + sigma), in fp64
    (reference :38-55; logit(t) = ln sigma)."""

    def __init__(self, m: float = 0.0, s: float = 1.0):
        self.m, self.s = m, s

    def __call__(self, sigma: Tensor) -> Tensor:
        sigma = sigma.to(torch.float64)
        t = sigma / (1.0 + sigma)
        return 1 / (1 - t) ** 2 * _logit_normal_density(t, torch.log(sigma), self.m, self.s)


class RectifiedFlowComfyWeighting(DenoiserWeighting):
    """the same with sigma = t (reference :58-75)"""

    def __init__(self, m: float = 0.0, s: float = 1.0):
        self.m, self.s = m, s

    def __call__(self, sigma: Tensor) -> Tensor:
        t = sigma.to(torch.float64)
        return 1 / (1 - t) ** 2 * _logit_normal_density(t, torch.log(t / (1 - t)), self.m, self.s)


class MinSNRGammaModifier(DenoiserWeighting):
    """min-SNR-gamma on top of another weighting (reference :78-101): w * min(snr, gamma) / snr  (/(snr + 1) for v-prediction),
    snr = sigma^-2."""

    def __init__(self, weighting: DenoiserWeighting, gamma: float = 5, v_pred: bool = False):
        self.weighting, self.gamma, self.v_pred = weighting, gamma, v_pred

    def __call__(self, sigma: Tensor) -> Tensor:
        snr = 1.0 / sigma**2
        capped = torch.min(snr, torch.full_like(snr, self.gamma))
        return self.weighting(sigma) * (capped / (snr + 1.0) if self.v_pred else capped / snr)
