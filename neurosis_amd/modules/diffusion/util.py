"""Host-side helpers mirrored from neurosis.modules.diffusion.util / neurosis.utils.sgm."""
from __future__ import annotations

import torch
from torch import Tensor

from ... import ops
from ..attention import zero_module  # noqa: F401  (util.py:180-186)


def timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000, repeat_only: bool = False) -> Tensor:
    """util.py:152-177 on the HIP kernel ([cos | sin], bf16)."""
    if repeat_only:
        raise NotImplementedError("repeat_only timestep embedding is not on the SD/SDXL path")
    if dim % 2:
        raise NotImplementedError("odd embedding dims are not on the SD/SDXL path")
    return ops.timestep_embedding(timesteps, dim, float(max_period))


def make_beta_schedule(schedule: str, n_timestep: int, linear_start: float = 1e-4, linear_end: float = 2e-2) -> Tensor:
    """util.py:22-46, "linear" branch (the one LegacyDDPMDiscretization uses)."""
    if schedule != "linear":
        raise ValueError(f"unknown or unsupported schedule: {schedule}")
    return torch.linspace(linear_start**0.5, linear_end**0.5, n_timestep, dtype=torch.float64) ** 2


def append_zero(x: Tensor) -> Tensor:
    """utils/sgm.py:141-142."""
    return torch.cat([x, x.new_zeros([1])])


def append_dims(x: Tensor, ndim: int) -> Tensor:
    """utils/sgm.py:145-150."""
    add_dims = ndim - x.ndim
    if add_dims < 0:
        raise ValueError(f"can't extend tensor from {x.ndim} to {ndim} dimensions!")
    return x[(...,) + (None,) * add_dims]
