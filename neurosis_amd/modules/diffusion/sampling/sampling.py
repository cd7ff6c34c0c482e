"""Samplers with the reference's names and arguments (`neurosis.modules.diffusion.sampling.sampling`, :28-457).

Structure here: one driver loop (`BaseDiffusionSampler.__call__`) walks the sigma table and asks the subclass to `advance`
one interval; subclasses keep the reference's extension points (`sampler_step`, `possible_correction_step`, `get_variables`,
`get_mult`, ...).  The sigma table stays on the host (as in the reference, whose `discretization(num_steps)` defaults to
device="cpu"), so every data-independent branch -- churn window, "is the next level zero" -- is decided from python floats
and the loop never waits for the GPU: the CPU runs ahead queueing the next UNet forward while the current one executes.

When the denoiser handed in is a `FusedDenoiser` (sampling/fused.py) the latents go through the nk_sample_* kernels:
guidance + c_skip/c_out scaling (+ the Euler update for EulerEDMSampler) in one launch, CFG batch duplication + c_in scaling
in another.  Any other callable gets the generic torch path, which is the reference's arithmetic op for op.
"""
from __future__ import annotations

import logging
from typing import Optional

import torch
from torch import Tensor

from ...guidance import Guider, IdentityGuider
from ..discretization import Discretization, RectifiedFlowComfyDiscretization
from .utils import _per_sample, get_ancestral_step, linear_multistep_coeff, to_d, to_neg_log_sigma, to_sigma

logger = logging.getLogger(__name__)

__all__ = [
    "AncestralSampler", "BaseDiffusionSampler", "DPMPP2MSampler", "DPMPP2SAncestralSampler", "EDMSampler", "EulerAncestralSampler",
    "EulerEDMSampler", "HeunEDMSampler", "LinearMultistepSampler", "SingleStepDiffusionSampler",
]

ZERO_LEVEL = 1e-14     # "all noise levels are 0" threshold of the reference's early-outs


class BaseDiffusionSampler:
    def __init__(self, discretization: Discretization, guider: Optional[Guider] = None, num_steps: Optional[int] = None, verbose: bool = False,
                 device="cuda", rf_safeguard: bool = False):
        self.discretization = discretization
        self.guider = guider if guider is not None else IdentityGuider()
        self.num_steps = num_steps
        self.verbose = verbose
        self.device = torch.device(device)
        self.rf_safeguard = rf_safeguard
        self._comfy_rf = isinstance(discretization, RectifiedFlowComfyDiscretization)
        if rf_safeguard and not self._comfy_rf:
            logger.warning("RF safeguard is only available for ComfyRF! Continuing without it.")
        self._host_levels: Optional[tuple] = None     # (sigma_i, sigma_{i+1}) as floats while the driver loop runs

    # -- set-up ------------------------------------------------------------------------------------
    def prepare_sampling_loop(self, x: Tensor, cond, uc=None, num_steps: Optional[int] = None):
        steps = self.num_steps if num_steps is None else num_steps
        if steps is None:
            raise ValueError(f"Step count must be set at init or call time! {self.num_steps=}")
        sigmas = self.discretization(steps)
        # unit-variance noise -> the scale of the first level, in place (rectified flow with sigma = t: x_t = (1 - t) x0 + t eps)
        x *= sigmas[0] if self._comfy_rf else torch.sqrt(1.0 + sigmas[0] ** 2.0)
        return x, x.new_ones([x.shape[0]]), sigmas, len(sigmas), cond, cond if uc is None else uc

    def get_sigma_gen(self, num_sigmas: int):
        steps = range(num_sigmas - 1)
        if not self.verbose:
            return steps
        from tqdm import tqdm

        logger.info("sampler %s / discretization %s / guider %s", *(type(o).__name__ for o in (self, self.discretization, self.guider)))
        return tqdm(steps, total=num_sigmas, desc=f"Sampling with {type(self).__name__} for {num_sigmas} steps")

    # -- one guided denoiser evaluation ------------------------------------------------------------
    def denoise(self, x: Tensor, denoiser, sigma: Tensor, cond, uc) -> Tensor:
        fused = getattr(denoiser, "guided", None)
        if fused is not None and denoiser.supports(x, self.guider, cond):
            return fused(x, sigma, cond, uc, self.guider)
        denoised = self.guider(denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc)), sigma)
        if self._comfy_rf and self.rf_safeguard:
            # reference :78-89: samples whose implied x0 has a standard deviation outside [0.5, 1.5] are renormalised
            x0 = denoised / (1.0 - _per_sample(sigma, x))
            std = x0.std(dim=tuple(range(1, denoised.dim())))
            off = (std < 0.5) | (std > 1.5)
            denoised[off] /= std[off].view(-1, *[1] * (denoised.dim() - 1))
        return denoised

    def _level_is_zero(self, level: Tensor, which: int) -> bool:
        """reference: torch.sum(level) < 1e-14.  Inside the driver loop the answer comes from the host table (no sync)."""
        if self._host_levels is not None and which is not None:
            return self._host_levels[which] * level.numel() < ZERO_LEVEL
        return bool(torch.sum(level) < ZERO_LEVEL)

    # -- driver ------------------------------------------------------------------------------------
    def begin(self, sigmas: Tensor):
        """per-run state of the subclass"""
        return None

    def advance(self, i: int, x: Tensor, sigmas: Tensor, s_in: Tensor, denoiser, cond, uc, state):
        raise NotImplementedError("Abstract base class was called ;_;")

    def __call__(self, denoiser, x: Tensor, cond, uc=None, num_steps: Optional[int] = None, **kwargs) -> Tensor:
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        host = [float(s) for s in sigmas]
        state = self.begin(sigmas)
        try:
            for i in self.get_sigma_gen(num_sigmas):
                self._host_levels = (host[i], host[i + 1])
                x = self.advance(i, x, sigmas, s_in, denoiser, cond, uc, state)
        finally:
            self._host_levels = None
        return x


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, *args, **kwargs):
        raise NotImplementedError("Abstract base class was called ;_;")

    def euler_step(self, x: Tensor, d: Tensor, dt: Tensor) -> Tensor:
        return x + dt * d

    def advance(self, i, x, sigmas, s_in, denoiser, cond, uc, state):
        return self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc)


# ---------------------------------------------------------------------------------------------------
# EDM (Karras et al. 2022, algorithm 2): optional churn, Euler predictor, optional 2nd-order corrector
# ---------------------------------------------------------------------------------------------------
class EDMSampler(SingleStepDiffusionSampler):
    def __init__(self, s_churn: float = 0.0, s_tmin: float = 0.0, s_tmax: float = float("inf"), s_noise: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise

    fuses_euler = False      # EulerEDMSampler: the whole step is one kernel after the network

    def sampler_step(self, sigma: Tensor, next_sigma: Tensor, denoiser, x: Tensor, cond, uc=None, gamma: float = 0.0) -> Tensor:
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            bump = _per_sample(sigma_hat**2 - sigma**2, x) ** 0.5
            x = x + torch.randn_like(x) * self.s_noise * bump
        if self._plain_euler() and getattr(denoiser, "euler", None) is not None and denoiser.supports(x, self.guider, cond):
            return denoiser.euler(x, sigma_hat, next_sigma, cond, uc, self.guider)
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
        d = to_d(x, sigma_hat, denoised)
        dt = _per_sample(next_sigma - sigma_hat, x)
        return self.possible_correction_step(self.euler_step(x, d, dt), x, d, dt, next_sigma, denoiser, cond, uc)

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        raise NotImplementedError("Abstract base class was called ;_;")

    def _plain_euler(self) -> bool:
        """the fused kernel computes exactly `x + dt * d` with no corrector: only when neither hook has been overridden"""
        cls = type(self)
        return self.fuses_euler and cls.euler_step is SingleStepDiffusionSampler.euler_step and \
            cls.possible_correction_step is EulerEDMSampler.possible_correction_step

    def advance(self, i, x, sigmas, s_in, denoiser, cond, uc, state):
        level = self._host_levels[0]
        in_window = self.s_tmin <= level <= self.s_tmax
        gamma = min(self.s_churn / (len(sigmas) - 1), 2**0.5 - 1) if in_window else 0.0
        return self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, gamma)


class EulerEDMSampler(EDMSampler):
    fuses_euler = True

    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        return euler_step


class HeunEDMSampler(EDMSampler):
    def possible_correction_step(self, euler_step, x, d, dt, next_sigma, denoiser, cond, uc):
        if self._level_is_zero(next_sigma, 1):
            return euler_step                      # last interval: no second network evaluation
        d_next = to_d(euler_step, next_sigma, self.denoise(euler_step, denoiser, next_sigma, cond, uc))
        trapezoid = x + 0.5 * (d + d_next) * dt
        return torch.where(_per_sample(next_sigma, x) > 0.0, trapezoid, euler_step)


# ---------------------------------------------------------------------------------------------------
# ancestral samplers: step down to sigma_down deterministically, then add sigma_up of fresh noise
# ---------------------------------------------------------------------------------------------------
class AncestralSampler(SingleStepDiffusionSampler):
    def __init__(self, eta: float = 1.0, s_noise: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eta, self.s_noise = eta, s_noise
        self.noise_sampler = lambda x: torch.randn_like(x)

    def ancestral_euler_step(self, x: Tensor, denoised: Tensor, sigma: Tensor, sigma_down: Tensor) -> Tensor:
        return self.euler_step(x, to_d(x, sigma, denoised), _per_sample(sigma_down - sigma, x))

    def ancestral_step(self, x: Tensor, sigma, next_sigma: Tensor, sigma_up) -> Tensor:
        up = _per_sample(sigma_up, x) if torch.is_tensor(sigma_up) else sigma_up
        noised = x + self.noise_sampler(x) * self.s_noise * up
        return torch.where(_per_sample(next_sigma, x) > 0.0, noised, x)


class EulerAncestralSampler(AncestralSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None):
        sigma_down, sigma_up = get_ancestral_step(sigma, next_sigma, eta=self.eta)
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        x = self.ancestral_euler_step(x, denoised, sigma, sigma_down)
        return self.ancestral_step(x, sigma, next_sigma, sigma_up)


class DPMPP2SAncestralSampler(AncestralSampler):
    """DPM-Solver++(2S): a midpoint evaluation in -log(sigma) time"""

    def get_variables(self, sigma, sigma_down):
        t, t_next = to_neg_log_sigma(sigma), to_neg_log_sigma(sigma_down)
        h = t_next - t
        return h, t + 0.5 * h, t, t_next

    def get_mult(self, h, s, t, t_next):
        return to_sigma(s) / to_sigma(t), torch.expm1(-0.5 * h), to_sigma(t_next) / to_sigma(t), torch.expm1(-h)

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, **kwargs):
        sigma_down, sigma_up = get_ancestral_step(sigma, next_sigma, eta=self.eta)
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        stepped = self.ancestral_euler_step(x, denoised, sigma, sigma_down)
        # for eta <= 1, sigma_down is 0 exactly when the next level is 0 (get_ancestral_step), so the host table answers this too
        if not self._level_is_zero(sigma_down, 1 if self.eta <= 1.0 else None):
            h, s, t, t_next = self.get_variables(sigma, sigma_down)
            m_mid, m_mid_d, m_end, m_end_d = (_per_sample(m, x) for m in self.get_mult(h, s, t, t_next))
            midpoint = m_mid * x - m_mid_d * denoised
            denoised_mid = self.denoise(midpoint, denoiser, to_sigma(s), cond, uc)
            second_order = m_end * x - m_end_d * denoised_mid
            stepped = torch.where(_per_sample(sigma_down, x) > 0.0, second_order, stepped)
        return self.ancestral_step(stepped, sigma, next_sigma, sigma_up)


# ---------------------------------------------------------------------------------------------------
# multistep samplers (carry history between intervals)
# ---------------------------------------------------------------------------------------------------
class LinearMultistepSampler(BaseDiffusionSampler):
    def __init__(self, order: int = 4, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.order = order

    def begin(self, sigmas: Tensor):
        return {"derivatives": [], "nodes": sigmas.detach().cpu().numpy()}

    def advance(self, i, x, sigmas, s_in, denoiser, cond, uc, state):
        sigma = s_in * sigmas[i]
        history = state["derivatives"]
        history.append(to_d(x, sigma, self.denoise(x, denoiser, sigma, cond, uc)))
        del history[: max(0, len(history) - self.order)]
        order = min(i + 1, self.order)
        update = sum(linear_multistep_coeff(order, state["nodes"], i, j) * d for j, d in zip(range(order), reversed(history)))
        return x + update


class DPMPP2MSampler(BaseDiffusionSampler):
    """DPM-Solver++(2M): reuses the previous interval's denoised estimate for the second-order term"""

    def get_variables(self, sigma, next_sigma, previous_sigma=None):
        t, t_next = to_neg_log_sigma(sigma), to_neg_log_sigma(next_sigma)
        h = t_next - t
        r = None if previous_sigma is None else (t - to_neg_log_sigma(previous_sigma)) / h
        return h, r, t, t_next

    def get_mult(self, h, r, t, t_next, previous_sigma):
        first_order = (to_sigma(t_next) / to_sigma(t), torch.expm1(-h))
        if previous_sigma is None:
            return first_order
        return first_order + (1 + 1 / (2 * r), 1 / (2 * r))

    def sampler_step(self, old_denoised, previous_sigma, sigma, next_sigma, denoiser, x, cond, uc=None):
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        h, r, t, t_next = self.get_variables(sigma, next_sigma, previous_sigma)
        mult = [_per_sample(m, x) for m in self.get_mult(h, r, t, t_next, previous_sigma)]
        x_standard = mult[0] * x - mult[1] * denoised
        if old_denoised is None or self._level_is_zero(next_sigma, 1):
            return x_standard, denoised            # first interval, or stepping to sigma = 0
        extrapolated = mult[2] * denoised - mult[3] * old_denoised
        x_advanced = mult[0] * x - mult[1] * extrapolated
        return torch.where(_per_sample(next_sigma, x) > 0.0, x_advanced, x_standard), denoised

    def begin(self, sigmas: Tensor):
        return {"old_denoised": None}

    def advance(self, i, x, sigmas, s_in, denoiser, cond, uc, state):
        previous = None if i == 0 else s_in * sigmas[i - 1]
        x, state["old_denoised"] = self.sampler_step(state["old_denoised"], previous, s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc=uc)
        return x
