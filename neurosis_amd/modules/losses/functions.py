"""Discriminator losses with the reference's names (`neurosis.modules.losses.functions`, :21-62).  The logits of a PatchGAN
are a few thousand values: plain torch arithmetic on the device; `with_grad` additionally returns d loss / d logits, which is
what the explicit backward of the discriminator consumes."""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import functional as F


class _DiscLoss(nn.Module):
    def __init__(self, weight: float = 1.0, start_step: int = 0):
        super().__init__()
        self.weight, self.start_step = weight, start_step

    def _terms(self, real: Tensor, fake: Tensor):
        raise NotImplementedError

    def with_grad(self, real: Tensor, fake: Tensor, global_step: int = -1):
        """(loss, d_loss/d_real, d_loss/d_fake)"""
        if self.start_step > 0 and global_step < self.start_step:
            return torch.zeros(1, device=real.device), torch.zeros_like(real), torch.zeros_like(fake)
        (l_real, g_real), (l_fake, g_fake) = self._terms(real.float(), fake.float())
        scale = 0.5 * self.weight
        return (l_real.mean() + l_fake.mean()) * scale, g_real * (scale / real.numel()), g_fake * (scale / fake.numel())

    def forward(self, real: Tensor, fake: Tensor, global_step: int = -1) -> Tensor:
        return self.with_grad(real, fake, global_step)[0]


class HingeDiscLoss(_DiscLoss):
    """0.5 * (mean relu(1 - real) + mean relu(1 + fake))"""

    def _terms(self, real, fake):
        return (F.relu(1.0 - real), -(real < 1.0).float()), (F.relu(1.0 + fake), (fake > -1.0).float())


class VanillaDiscLoss(_DiscLoss):
    """0.5 * (mean softplus(-real) + mean softplus(fake))"""

    def _terms(self, real, fake):
        return (F.softplus(-real), -torch.sigmoid(-real)), (F.softplus(fake), torch.sigmoid(fake))


def get_discr_loss_fn(kind: str = "hinge", weight: float = 1.0, start_step: int = 0):
    if kind == "hinge":
        return HingeDiscLoss(weight, start_step)
    if kind == "vanilla":
        return VanillaDiscLoss(weight, start_step)
    raise ValueError(f"Unknown discriminator loss: {kind}")
