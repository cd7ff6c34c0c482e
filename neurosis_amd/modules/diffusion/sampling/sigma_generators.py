"""Training-time sigma samplers (host side): mirror of neurosis.modules.diffusion.sampling.sigma_generators."""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from typing import Optional

import torch
from torch import Tensor

from ..discretization import Discretization


__all__ = ["CosineScheduleSigmaGenerator", "DiscreteSigmaGenerator", "EDMSigmaGenerator", "InjectedSigmaGenerator", "RectifiedFlowComfySigmaGenerator",
           "RectifiedFlowSigmaGenerator", "SigmaGenerator", "TanScheduleSigmaGenerator"]


class SigmaGenerator(ABC):
    @abstractmethod
    def __call__(self, n_samples: int, t: Optional[Tensor] = None): ...


class EDMSigmaGenerator(SigmaGenerator):
    """sigma_generators.py:17-35 (treats the uniform t it is handed as if it were normal -- reproduced)."""

    def __init__(self, p_mean: float = -1.2, p_std: float = 1.2, scale: float = 2.0):
        self.p_mean, self.p_std, self.scale = p_mean, p_std, scale

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        t = t.to(torch.float32) if t is not None else torch.randn((n_samples,), dtype=torch.float32)
        return (self.p_mean + self.p_std * t).exp() * self.scale


class DiscreteSigmaGenerator(SigmaGenerator):
    """sigma_generators.py:38-57.  With the loss's t ~ U[0,1) this always returns table[0] (SURVEY quirk Q3:
    0.0 for flip=True); reproduced as is -- use InjectedSigmaGenerator or EDMSigmaGenerator for real training."""

    def __init__(self, discretization: Discretization, num_idx: int = 1000, do_append_zero: bool = True, flip: bool = True):
        self.num_idx = num_idx
        self.sigmas = discretization(num_idx, do_append_zero=do_append_zero, flip=flip)

    def idx_to_sigma(self, idx) -> Tensor:
        return self.sigmas[idx]

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        idx = torch.clamp(t.long(), 0, self.num_idx - 1) if t is not None else torch.randint(0, self.num_idx, (n_samples,))
        return self.idx_to_sigma(idx)


def _uniform64(n_samples: int, t: Optional[Tensor]) -> Tensor:
    return torch.rand((n_samples,), dtype=torch.float64) if t is None else t.to(torch.float64)


class CosineScheduleSigmaGenerator(SigmaGenerator):
    """sigma from the cosine variance schedule (reference :60-90): var(t) = cos^2(pi/2 (s + t) / (1 + s)) / cos^2(pi/2 s / (1 + s)),
    squeezed into [1e-4, 1], logSNR = ln(var / (1 - var)) (+ 2 ln(1 / shift)), sigma = sigma_data * exp(-logSNR / 2)."""

    def __init__(self, s: float = 0.008, sigma_data: float = 1.0):
        self.s = torch.tensor([s])
        self.sigma_data = sigma_data
        self.min_var = torch.cos(self.s / (1 + self.s) * torch.pi * 0.5) ** 2

    def __call__(self, n_samples: int, t: Optional[Tensor] = None, shift: int = 1, return_logSNR: bool = False):
        if t is None:
            t = (1 - torch.rand(n_samples)).add(0.001).clamp(0.001, 1.0)
        s, floor = self.s.to(t.device), self.min_var.to(t.device)
        var = 0.0001 + 0.9999 * (torch.cos((s + t) / (1 + s) * torch.pi * 0.5).clamp(0, 1) ** 2 / floor)
        log_snr = (var / (1 - var)).log()
        if shift != 1:
            log_snr += 2 * math.log(1 / shift)
        return log_snr if return_logSNR else torch.exp(-log_snr / 2) * self.sigma_data


class TanScheduleSigmaGenerator(SigmaGenerator):
    """sigma = scale * tan(pi/2 t), the angle clipped away from 0 and pi/2 (reference :93-119; fp64 inside)"""

    def __init__(self, start_shift: float = 0.001, end_shift: float = 0.001, scale: float = 1.0, clip: bool = True):
        self.start_shift, self.end_shift, self.scale, self.clip = start_shift, end_shift, scale, clip

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        half_pi = torch.acos(torch.zeros(1, dtype=torch.float64))
        angle = half_pi * _uniform64(n_samples, t)
        if self.clip:
            angle = angle.clip(torch.tensor([self.start_shift], dtype=torch.float64), half_pi - self.end_shift)
        return torch.tan(angle).mul(self.scale).to(torch.float32)


class RectifiedFlowSigmaGenerator(SigmaGenerator):
    """t clipped to [start_shift, 1 - end_shift], sigma = t / (1 - t) (reference :122-143)"""

    as_ratio = True

    def __init__(self, start_shift: float = 0.0, end_shift: float = 0.001, clip: bool = True):
        self.start_shift, self.end_shift, self.clip = start_shift, end_shift, clip

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        t = _uniform64(n_samples, t)
        if self.clip:
            t = t.clip(self.start_shift, 1 - self.end_shift)
        return (t / (1 - t) if self.as_ratio else t).to(torch.float32)


class RectifiedFlowComfySigmaGenerator(RectifiedFlowSigmaGenerator):
    """the same with sigma = t (reference :146-166)"""

    as_ratio = False


class InjectedSigmaGenerator(SigmaGenerator):
    """The working stand-in SURVEY section 8(d) prescribes for benchmarks and parity tests:
    sigma = exp(p_mean + p_std * n), n ~ N(0,1), clipped to the discretisation's range."""

    def __init__(self, sigma_min: float = 0.0292, sigma_max: float = 14.6146, p_mean: float = -1.2, p_std: float = 1.2, generator: Optional[torch.Generator] = None):
        self.sigma_min, self.sigma_max, self.p_mean, self.p_std, self.generator = sigma_min, sigma_max, p_mean, p_std, generator

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        n = torch.randn((n_samples,), dtype=torch.float32, generator=self.generator)
        return (self.p_mean + self.p_std * n).exp().clamp(self.sigma_min, self.sigma_max)
