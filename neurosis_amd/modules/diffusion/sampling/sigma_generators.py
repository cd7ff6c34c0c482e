"""Training-time sigma samplers (host side): mirror of neurosis.modules.diffusion.sampling.sigma_generators."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch
from torch import Tensor

from ..discretization import Discretization


__all__ = ["DiscreteSigmaGenerator", "EDMSigmaGenerator", "InjectedSigmaGenerator", "SigmaGenerator"]


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


class InjectedSigmaGenerator(SigmaGenerator):
    """The working stand-in SURVEY section 8(d) prescribes for benchmarks and parity tests:
    sigma = exp(p_mean + p_std * n), n ~ N(0,1), clipped to the discretisation's range."""

    def __init__(self, sigma_min: float = 0.0292, sigma_max: float = 14.6146, p_mean: float = -1.2, p_std: float = 1.2, generator: Optional[torch.Generator] = None):
        self.sigma_min, self.sigma_max, self.p_mean, self.p_std, self.generator = sigma_min, sigma_max, p_mean, p_std, generator

    def __call__(self, n_samples: int, t: Optional[Tensor] = None):
        n = torch.randn((n_samples,), dtype=torch.float32, generator=self.generator)
        return (self.p_mean + self.p_std * n).exp().clamp(self.sigma_min, self.sigma_max)
