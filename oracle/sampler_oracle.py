"""CPU oracle for the sampling loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

A flat, class-free restatement of the reference's samplers (modules/diffusion/sampling/sampling.py, paths relative to
/root/reference/src/neurosis/): every function takes `denoise(x, sigma[B]) -> denoised`, the already-guided denoiser, and
the host sigma table (descending, final 0).  Pinned by tests/golden/sampler_analytic.pt and sampler_unet_tiny.pt, captured
from the reference's own sampler classes by tests/golden/make_golden.py.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
from torch import Tensor

Denoise = Callable[[Tensor, Tensor], Tensor]


def _b(v: Tensor, x: Tensor) -> Tensor:
    return v.reshape(-1, *([1] * (x.ndim - 1)))


def cfg_denoise(denoiser, scale: Optional[float], cond: dict, uc: dict) -> Denoise:
    """BaseDiffusionSampler.denoise (sampling.py:73-76) with VanillaCFG / IdentityGuider (modules/guidance.py:17-47):
    the batch is stacked [uncond | cond], evaluated once, and recombined as u + scale * (c - u)."""
    def run(x: Tensor, sigma: Tensor) -> Tensor:
        if scale is None:
            return denoiser(x, sigma, cond)
        stacked = {k: torch.cat((uc[k], cond[k]), 0) for k in cond}
        u, c = denoiser(torch.cat([x, x]), torch.cat([sigma, sigma]), stacked).chunk(2)
        return u + scale * (c - u)
    return run


def start(x: Tensor, sigmas: Tensor) -> Tensor:
    """prepare_sampling_loop, sampling.py:51-71"""
    return x * torch.sqrt(1.0 + sigmas[0] ** 2.0)


def edm(denoise: Denoise, x: Tensor, sigmas: Tensor, heun: bool = False, s_churn: float = 0.0, s_tmin: float = 0.0, s_tmax: float = float("inf"),
        s_noise: float = 1.0, record: Optional[list] = None) -> Tensor:
    """EDMSampler.__call__/sampler_step with the Euler (:313-316) or Heun (:319-333) correction, sampling.py:140-208."""
    x = start(x, sigmas)
    ones = x.new_ones(x.shape[0])
    n = len(sigmas)
    for i in range(n - 1):
        gamma = min(s_churn / (n - 1), 2 ** 0.5 - 1) if s_tmin <= float(sigmas[i]) <= s_tmax else 0.0
        sigma, nxt = ones * sigmas[i], ones * sigmas[i + 1]
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            x = x + torch.randn_like(x) * s_noise * _b(sigma_hat ** 2 - sigma ** 2, x) ** 0.5
        d = (x - denoise(x, sigma_hat)) / _b(sigma_hat, x)
        dt = _b(nxt - sigma_hat, x)
        euler = x + dt * d
        if heun and float(nxt.sum()) >= 1e-14:
            d2 = (euler - denoise(euler, nxt)) / _b(nxt, x)
            x = torch.where(_b(nxt, x) > 0.0, x + (d + d2) / 2.0 * dt, euler)
        else:
            x = euler
        if record is not None:
            record.append(x.clone())
    return x


def ancestral_levels(sigma_from: Tensor, sigma_to: Tensor, eta: float):
    """get_ancestral_step, sampling/utils.py:36-46"""
    up = torch.min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - up ** 2) ** 0.5, up


def euler_ancestral(denoise: Denoise, x: Tensor, sigmas: Tensor, noise: Callable[[Tensor], Tensor], eta: float = 1.0, s_noise: float = 1.0,
                    dpmpp2s: bool = False) -> Tensor:
    """EulerAncestralSampler (:336-343) / DPMPP2SAncestralSampler (:346-384) on AncestralSampler (:211-269)."""
    x = start(x, sigmas)
    ones = x.new_ones(x.shape[0])
    for i in range(len(sigmas) - 1):
        sigma, nxt = ones * sigmas[i], ones * sigmas[i + 1]
        down, up = ancestral_levels(sigma, nxt, eta)
        denoised = denoise(x, sigma)
        stepped = x + _b(down - sigma, x) * ((x - denoised) / _b(sigma, x))
        if dpmpp2s and float(down.sum()) >= 1e-14:
            t, t_next = -sigma.log(), -down.log()
            h = t_next - t
            s = t + 0.5 * h
            sig = lambda v: (-v).exp()  # noqa: E731
            x2 = _b(sig(s) / sig(t), x) * x - _b((-0.5 * h).expm1(), x) * denoised
            denoised2 = denoise(x2, sig(s))
            second = _b(sig(t_next) / sig(t), x) * x - _b((-h).expm1(), x) * denoised2
            stepped = torch.where(_b(down, x) > 0.0, second, stepped)
        x = torch.where(_b(nxt, x) > 0.0, stepped + noise(stepped) * s_noise * _b(up, x), stepped)
    return x


def dpmpp2m(denoise: Denoise, x: Tensor, sigmas: Tensor) -> Tensor:
    """DPMPP2MSampler, sampling.py:387-457."""
    x = start(x, sigmas)
    ones = x.new_ones(x.shape[0])
    old = None
    for i in range(len(sigmas) - 1):
        sigma, nxt = ones * sigmas[i], ones * sigmas[i + 1]
        denoised = denoise(x, sigma)
        t, t_next = -sigma.log(), -nxt.log()
        h = t_next - t
        ratio, decay = _b((-t_next).exp() / (-t).exp(), x), _b((-h).expm1(), x)
        standard = ratio * x - decay * denoised
        if old is None or float(nxt.sum()) < 1e-14:
            x = standard
        else:
            r = (t - (-(ones * sigmas[i - 1]).log())) / h
            blend = _b(1 + 1 / (2 * r), x) * denoised - _b(1 / (2 * r), x) * old
            x = torch.where(_b(nxt, x) > 0.0, ratio * x - decay * blend, standard)
        old = denoised
    return x


def lms(denoise: Denoise, x: Tensor, sigmas: Tensor, order: int = 4) -> Tensor:
    """LinearMultistepSampler, sampling.py:272-310 with linear_multistep_coeff (sampling/utils.py:18-33): Adams-Bashforth
    weights from integrating the Lagrange basis over each sigma interval (scipy quad, epsrel 1e-4)."""
    from scipy import integrate

    x = start(x, sigmas)
    ones = x.new_ones(x.shape[0])
    t = sigmas.detach().cpu().numpy()
    ds: list = []
    for i in range(len(sigmas) - 1):
        sigma = ones * sigmas[i]
        ds.append((x - denoise(x, sigma)) / _b(sigma, x))
        ds = ds[-order:]
        cur = min(i + 1, order)

        def coeff(j: int) -> float:
            def fn(tau):
                prod = 1.0
                for k in range(cur):
                    if k != j:
                        prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
                return prod
            return integrate.quad(fn, t[i], t[i + 1], epsrel=1e-4)[0]

        x = x + sum(coeff(j) * d for j, d in zip(range(cur), reversed(ds)))
    return x
