"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Adafactor update (never imported by the product path).

Follows `neurosis/optimizers/adafactor.py`: `_get_lr` :133-147, `_rms` :150-151, `_approx_sq_grad` :154-159 and the body
of `step` :176-255 (beta1 = None branch).  Pinned by `tests/golden/adafactor_steps.pt`, produced by running the reference
class itself (tests/golden/make_golden.py::adafactor_case).  Plain torch fp32 on CPU, one tensor at a time.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor


def new_state(p: Tensor) -> dict:
    """adafactor.py:199-213"""
    st = {"step": 0, "RMS": 0.0}
    if p.dim() >= 2:
        st["exp_avg_sq_row"] = torch.zeros(p.shape[:-1])
        st["exp_avg_sq_col"] = torch.zeros(p.shape[:-2] + p.shape[-1:])
    else:
        st["exp_avg_sq"] = torch.zeros_like(p)
    return st


def rel_lr(step: int, rms: float, lr: Optional[float], eps2: float, scale_parameter: bool, relative_step: bool, warmup_init: bool) -> float:
    """adafactor.py:133-147"""
    rel = lr
    if relative_step:
        min_step = 1e-6 * step if warmup_init else 1e-2
        rel = min(min_step, 1.0 / math.sqrt(step))
    scale = max(eps2, rms) if scale_parameter else 1.0
    return scale * rel


def step_tensor(p: Tensor, grad: Tensor, st: dict, lr: Optional[float] = None, eps=(1e-30, 1e-3), clip_threshold: float = 1.0,
                decay_rate: float = -0.8, weight_decay: float = 0.0, scale_parameter: bool = True, relative_step: bool = True,
                warmup_init: bool = False) -> float:
    """One in-place update of `p` (fp32); returns the lr used.  adafactor.py:226-253"""
    st["step"] += 1
    st["RMS"] = float(p.norm(2) / (p.numel() ** 0.5))
    lr_t = rel_lr(st["step"], st["RMS"], lr, eps[1], scale_parameter, relative_step, warmup_init)
    beta2t = 1.0 - math.pow(st["step"], decay_rate)
    update = grad ** 2 + eps[0]
    if p.dim() >= 2:
        row, col = st["exp_avg_sq_row"], st["exp_avg_sq_col"]
        row.mul_(beta2t).add_(update.mean(dim=-1), alpha=1.0 - beta2t)
        col.mul_(beta2t).add_(update.mean(dim=-2), alpha=1.0 - beta2t)
        r_factor = (row / row.mean(dim=-1, keepdim=True)).rsqrt().unsqueeze(-1)
        c_factor = col.unsqueeze(-2).rsqrt()
        update = r_factor * c_factor * grad
    else:
        v = st["exp_avg_sq"]
        v.mul_(beta2t).add_(update, alpha=1.0 - beta2t)
        update = v.rsqrt() * grad
    rms_u = update.norm(2) / (update.numel() ** 0.5)
    update = update / (rms_u / clip_threshold).clamp(min=1.0)
    update = update * lr_t
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr_t)
    p.sub_(update)
    return lr_t
