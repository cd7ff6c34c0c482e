"""Guiders with the reference's names (`neurosis.modules.guidance`, :9-89).

A guider does two things around one denoiser call: `prepare_inputs` lays the batch out for the network (classifier-free
guidance: [unconditional | conditional] stacked on the batch axis) and `__call__` folds the stacked prediction back into one.
With `FusedDenoiser` (sampling/fused.py) both halves happen inside the nk_sample_* kernels instead; these classes then only
supply `rep` and `scale`.
"""
from __future__ import annotations

import torch
from torch import Tensor

STACKED_KEYS = ("vector", "crossattn", "concat")


def _stack_conditioning(c: dict, uc: dict, stacked=STACKED_KEYS) -> dict:
    out = {}
    for key, value in c.items():
        if key in stacked:
            out[key] = torch.cat((uc[key], value), 0)
        elif value != uc[key]:
            raise ValueError(f"Conditioning key {key} value mismatch between contexts!")
        else:
            out[key] = value
    return out


class Guider:
    rep = 1           # how many copies of the batch the network sees

    def __call__(self, x: Tensor, sigma) -> Tensor:
        raise NotImplementedError("Abstract base class was called ;_;")

    def prepare_inputs(self, x: Tensor, s, c: dict, uc: dict):
        raise NotImplementedError("Abstract base class was called ;_;")


class IdentityGuider(Guider):
    def __call__(self, x: Tensor, sigma) -> Tensor:
        return x

    def prepare_inputs(self, x: Tensor, s, c: dict, uc: dict):
        return x, s, dict(c)


class VanillaCFG(Guider):
    rep = 2

    def __init__(self, scale: float):
        self.scale = scale

    def __call__(self, x: Tensor, sigma) -> Tensor:
        uncond, cond = x.chunk(2)
        return uncond + self.scale * (cond - uncond)

    def prepare_inputs(self, x: Tensor, s: Tensor, c: dict, uc: dict):
        return torch.cat((x, x)), torch.cat((s, s)), _stack_conditioning(c, uc)


class LinearPredictionGuider(Guider):
    """per-frame guidance scale ramp (video models; kept for API completeness, not on the SDXL path)"""
    rep = 2

    def __init__(self, max_scale: float, num_frames: int, min_scale: float = 1.0, additional_cond_keys=()):
        self.min_scale, self.max_scale, self.num_frames = min_scale, max_scale, num_frames
        self.scale = torch.linspace(min_scale, max_scale, num_frames).unsqueeze(0)
        if isinstance(additional_cond_keys, str):
            additional_cond_keys = [additional_cond_keys]
        self.additional_cond_keys = list(additional_cond_keys)

    def __call__(self, x: Tensor, sigma) -> Tensor:
        uncond, cond = (h.reshape(-1, self.num_frames, *h.shape[1:]) for h in x.chunk(2))
        ramp = self.scale.to(x.device).reshape((1, self.num_frames) + (1,) * (uncond.ndim - 2))
        return (uncond + ramp * (cond - uncond)).flatten(0, 1)

    def prepare_inputs(self, x: Tensor, s: Tensor, c: dict, uc: dict):
        return torch.cat((x, x)), torch.cat((s, s)), _stack_conditioning(c, uc, STACKED_KEYS + tuple(self.additional_cond_keys))
