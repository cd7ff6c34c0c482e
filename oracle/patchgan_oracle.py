"""CPU oracle for the PatchGAN discriminator and its losses -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Functional fp32 restatement of `NLayerDiscriminator.forward` in training mode (modules/losses/patchgan/model.py:21-95; paths
relative to /root/reference/src/neurosis/) over a state_dict with the reference's keys, and of the hinge / vanilla
discriminator losses (modules/losses/functions.py:21-50).  Gradients come from torch autograd.  Pinned by
tests/golden/patchgan_tiny.pt (tests/golden/make_golden.py::discriminator_case)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def discriminator(sd: dict, x: Tensor, n_layers: int = 3, momentum: float = 0.1, eps: float = 1e-5, running: dict | None = None) -> Tensor:
    """layers.0 conv(4x4, s2, p1)+bias -> LeakyReLU(0.2); for n = 1..n_layers: conv(4x4, stride 2 except the last, no bias) ->
    BatchNorm2d (batch statistics; `running` receives the updated running_mean / running_var) -> LeakyReLU(0.2); final
    conv(4x4, s1, p1)+bias to one channel."""
    h = F.leaky_relu(F.conv2d(x, sd["layers.0.weight"], sd["layers.0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        h = F.conv2d(h, sd[f"layers.{idx}.weight"], None, stride=2 if n < n_layers else 1, padding=1)
        bn = f"layers.{idx + 1}"
        mean = h.mean(dim=(0, 2, 3))
        var = h.var(dim=(0, 2, 3), unbiased=False)
        if running is not None:
            count = h.numel() / h.shape[1]
            src = running if bn + ".running_mean" in running else sd
            running[bn + ".running_mean"] = (1 - momentum) * src[bn + ".running_mean"] + momentum * mean.detach()
            running[bn + ".running_var"] = (1 - momentum) * src[bn + ".running_var"] + momentum * var.detach() * count / (count - 1)
        h = (h - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
        h = F.leaky_relu(h * sd[bn + ".weight"][None, :, None, None] + sd[bn + ".bias"][None, :, None, None], 0.2)
        idx += 3
    return F.conv2d(h, sd[f"layers.{idx}.weight"], sd[f"layers.{idx}.bias"], stride=1, padding=1)


def disc_loss(kind: str, real: Tensor, fake: Tensor) -> Tensor:
    if kind == "hinge":
        return 0.5 * (F.relu(1.0 - real).mean() + F.relu(1.0 + fake).mean())
    if kind == "vanilla":
        return 0.5 * (F.softplus(-real).mean() + F.softplus(fake).mean())
    raise ValueError(kind)


def generator_adversarial_loss(enc_sd: dict, dec_sd: dict, disc_sd: dict, dd: dict, x: Tensor, noise: Tensor, rec_weight: float = 1.0, logvar: float = 0.0,
                               disc_factor: float = 1.0, disc_weight: float = 1.0, n_layers: int = 3, lpips=None, perceptual_weight: float = 1.0):
    """The autoencoder's side of GeneralLPIPSWithDiscriminator with perceptual_weight = 0 (discriminator_loss.py:205-233,
    247-286), as its terms spell it out -- NOTE: as its engine calls it (weights = None) the reference's own forward raises in this
    branch (`if weights > 0` at :300) and the loss it builds at :281 is an un-reduced [B, C, H, W] tensor that manual_backward cannot
    take.  Given a tensor `weights` the forward does run: its nll_loss, g_loss and adaptive weight are captured in
    tests/golden/gan_generator_tiny (make_golden.py::gan_generator_case) and this function is checked against them
    (tests/test_patchgan_cpu.py).  The TOTAL follows the taming-transformers / generative-models formula the reference was reworked from:
        nll = sum(rec_weight * (x - xrec)^2 / exp(logvar) + logvar) / B ;  g = -mean(D(xrec))
        d_w = clamp(||grad_W nll|| / (||grad_W g|| + 1e-4), 0, 1e4) * disc_weight,  W = decoder.conv_out.weight
        loss = nll + disc_factor * d_w * g
    `dec_sd` tensors must require grad.  Returns (loss, nll, g, d_w, xrec)."""
    from oracle import sdxl_oracle as O

    moments = O.vae_moments(enc_sd, dd, x)
    mean, lv = torch.chunk(moments, 2, dim=1)
    z = mean + torch.exp(0.5 * lv.clamp(-30.0, 20.0)) * noise
    xrec = O.vae_decode(dec_sd, dd, z)
    p_rec = (x - xrec) ** 2 * rec_weight
    if lpips is not None:        # (trunk weights, lin weights): p_rec_loss = rec_weight * rec + perceptual_weight * p_loss, broadcast (:254-259)
        from oracle import lpips_oracle as LO

        p_rec = p_rec + perceptual_weight * LO.lpips(lpips[0], lpips[1], x, xrec)
    nll = (p_rec / float(torch.exp(torch.tensor(logvar))) + logvar).sum() / x.shape[0]
    g = -discriminator(disc_sd, xrec, n_layers).mean()
    last = dec_sd["conv_out.weight"]
    nll_grad = torch.autograd.grad(nll, last, retain_graph=True)[0]
    g_grad = torch.autograd.grad(g, last, retain_graph=True)[0]
    d_w = (nll_grad.norm() / (g_grad.norm() + 1e-4)).clamp(0.0, 1e4).detach() * disc_weight
    return nll + disc_factor * d_w * g, nll, g, d_w, xrec
