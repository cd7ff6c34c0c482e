"""The autoencoder-training loss CLASSES of the reference, with the reference's constructors, so that a config-5 YAML
(`loss: {class_path: neurosis.modules.autoencoding.losses.GeneralLPIPSWithDiscriminator, init_args: {...}}`) prefix-swaps like the
SDXL one does.

    GeneralLPIPSWithDiscriminator   /root/reference/src/neurosis/modules/autoencoding/losses/discriminator_loss.py:22-88
    AutoencoderLPIPSWithDiscr       /root/reference/src/neurosis/modules/autoencoding/losses/vae_lpips_discr.py:140-200

In the reference these modules own the LPIPS network, the PatchGAN discriminator and the output log-variance, and their `forward`
assembles the loss from torch ops under autograd.  Here they OWN THE SAME SUBMODULES UNDER THE SAME NAMES (so `loss.*` checkpoint
keys line up: `loss.discriminator.*` / `loss.discr.*`, `loss.perceptual_loss.*`, `loss.logvar`) and carry the configuration;
the arithmetic is the fused one of `neurosis_amd.models.autoencoder.AutoencodingEngine`, which reads its settings from the loss
object it is handed (`AutoencodingEngine(loss=<one of these>)`): nll with the learnable log-variance, LPIPS broadcast into the
reconstruction term, the adaptive generator weight from two last-layer weight-gradient norms, hinge / vanilla discriminator
losses with explicit logit gradients, alternating optimizers from `disc_start` on.  Calling the object directly raises: there is
no autograd graph through the HIP decoder for a stand-alone loss call to differentiate.

Options outside the SD / SDXL autoencoder recipes are refused loudly at construction (3-D inputs, input rescaling, R1
penalty, non-LPIPS perceptual types) rather than accepted and ignored.
"""
from __future__ import annotations

from sys import maxsize
from typing import Any, Iterator, Optional, Union

import torch
from torch import nn

from ...losses import LPIPS, NLayerDiscriminator, get_discr_loss_fn, weights_init

__all__ = ["AutoencoderLPIPSWithDiscr", "GeneralLPIPSWithDiscriminator"]


def _rec_type(kind) -> str:
    name = str(getattr(kind, "value", kind)).lower()
    if name in ("l2", "mse"):
        return "l2"
    if name in ("l1", "mae"):
        return "l1"
    raise ValueError(f"Unknown reconstruction loss type {kind}")


class _FusedLossConfig(nn.Module):
    """What AutoencodingEngine reads from a loss object (`engine_settings`)."""

    rec_loss_type: str
    rec_weight: float
    perceptual_weight: float
    disc_start: int
    disc_factor: float
    discriminator_weight: float
    disc_loss_name: str
    learn_logvar: bool

    def engine_settings(self) -> dict:
        return dict(rec_loss_type=self.rec_loss_type, rec_weight=self.rec_weight, perceptual_loss=self.perceptual_loss,
                    perceptual_weight=self.perceptual_weight, discriminator=self._discriminator_module(), disc_loss=self.disc_loss_name,
                    disc_start=self.disc_start, disc_factor=self.disc_factor, disc_weight=self.discriminator_weight,
                    logvar=self._logvar_parameter(), learn_logvar=self.learn_logvar, regularization_weights=dict(getattr(self, "regularization_weights", {}) or {}))

    def _discriminator_module(self) -> nn.Module:
        raise NotImplementedError

    def _logvar_parameter(self) -> Optional[nn.Parameter]:
        return None

    def forward(self, *args, **kwargs):
        raise RuntimeError(f"{type(self).__name__} configures the fused loss of neurosis_amd.models.autoencoder.AutoencodingEngine "
                           "(pass it as `loss=`); it is not a stand-alone autograd loss on this backend")


class GeneralLPIPSWithDiscriminator(_FusedLossConfig):
    """discriminator_loss.py:22-88 (constructor), :205-320 (what the engine computes)."""

    def __init__(self, disc_start: int, logvar_init: float = 0.0, disc_num_layers: int = 3, disc_in_channels: int = 3, disc_factor: float = 1.0,
                 disc_weight: float = 1.0, perceptual_weight: float = 1.0, disc_loss: str = "hinge", scale_input_to_tgt_size: bool = False, dims: int = 2,
                 learn_logvar: bool = False, rec_loss_type: str = "l2", rec_weight: float = 1.0,
                 regularization_weights: Union[None, dict[str, float]] = None, additional_log_keys: Optional[list[str]] = None,
                 discriminator_config: Optional[dict] = None, lpips_kwargs: Optional[dict] = None):
        super().__init__()
        if dims != 2:
            raise NotImplementedError("dims > 2 (video autoencoders) is outside the SD / SDXL path")
        if scale_input_to_tgt_size:
            raise NotImplementedError("scale_input_to_tgt_size is not built")
        if disc_loss not in ("hinge", "vanilla"):
            raise ValueError(f"disc_loss must be one of ['hinge', 'vanilla'], got {disc_loss}")
        self.dims, self.scale_input_to_tgt_size = dims, scale_input_to_tgt_size
        # `lpips_kwargs` is this package's addition: the reference constructs LPIPS() with its packaged weights; here the calibrated lin
        # weights are named explicitly (modules/losses/perceptual.py) or pretrained=False is passed for synthetic runs
        self.perceptual_loss = LPIPS(**(lpips_kwargs or {})).eval() if perceptual_weight > 0 else None
        self.perceptual_weight = float(perceptual_weight)
        self.logvar = nn.Parameter(torch.ones(size=()) * logvar_init, requires_grad=learn_logvar)
        self.learn_logvar = learn_logvar
        disc_kwargs = dict(input_nc=disc_in_channels, n_layers=disc_num_layers, use_actnorm=False)
        if discriminator_config is not None:
            disc_kwargs.update(discriminator_config)
        self.discriminator = NLayerDiscriminator(**disc_kwargs).apply(weights_init)
        self.disc_start = int(disc_start)
        self.disc_loss_name = disc_loss
        self.disc_loss = get_discr_loss_fn(disc_loss)
        self.disc_factor = float(disc_factor)
        self.discriminator_weight = float(disc_weight)
        self.rec_weight = float(rec_weight)
        self.rec_loss_type = _rec_type(rec_loss_type)
        self.regularization_weights = dict(regularization_weights or {})
        self.forward_keys = ["optimizer_idx", "global_step", "last_layer", "split", "regularization_log"]
        self.additional_log_keys = set(additional_log_keys or [])
        self.additional_log_keys.update(set(self.regularization_weights.keys()))

    def get_trainable_parameters(self) -> Iterator[nn.Parameter]:
        return self.discriminator.parameters()

    def get_trainable_autoencoder_parameters(self) -> Any:
        if self.learn_logvar:
            yield self.logvar
        yield from ()

    def _discriminator_module(self) -> nn.Module:
        return self.discriminator

    def _logvar_parameter(self) -> Optional[nn.Parameter]:
        return self.logvar


class AutoencoderLPIPSWithDiscr(_FusedLossConfig):
    """vae_lpips_discr.py:140-200: l1 / l2 reconstruction + LPIPS + PatchGAN with the discriminator under `discr`; no learnable
    log-variance (the nll reduces to the weighted reconstruction sum per sample)."""

    def __init__(self, recon_type="l1", recon_weight: float = 1.0, perceptual_type="lpips", perceptual_weight: float = 1.0, disc_start: int = -1,
                 disc_factor: float = 1.0, disc_weight: float = 1.0, disc_lambda_r1: float = 0.0, disc_loss="hinge", disc_kwargs: Optional[dict] = None,
                 resize_input: bool = False, resize_target: bool = False, extra_log_keys: Optional[list[str]] = None, lpips_kwargs: Optional[dict] = None):
        super().__init__()
        self.recon_type = recon_type
        self.rec_loss_type = _rec_type(recon_type)
        self.recon_weight = self.rec_weight = float(recon_weight)
        if str(getattr(perceptual_type, "value", perceptual_type)).lower() != "lpips":
            raise NotImplementedError(f"Perceptual loss {perceptual_type} not implemented")
        if resize_input or resize_target:
            raise NotImplementedError("resize_input / resize_target are not built")
        if disc_lambda_r1:
            raise NotImplementedError("the R1 gradient penalty (disc_lambda_r1) needs a double backward through the discriminator: not built")
        self.perceptual_loss = LPIPS(**(lpips_kwargs or {})).eval() if perceptual_weight > 0 else None
        self.perceptual_weight = float(perceptual_weight)
        self.disc_start = disc_start if disc_start > 0 else maxsize      # negative = never start (the reference's INT64_MAX hack)
        self.disc_factor, self.disc_weight, self.disc_lambda_r1 = float(disc_factor), float(disc_weight), float(disc_lambda_r1)
        self.discriminator_weight = self.disc_weight
        disc_config = dict(input_nc=3, n_layers=3, use_actnorm=False)
        if disc_kwargs is not None:
            disc_config.update(disc_kwargs)
        self.discr = NLayerDiscriminator(**disc_config).apply(weights_init)
        self.disc_loss_name = str(getattr(disc_loss, "value", disc_loss)).lower()
        if self.disc_loss_name not in ("hinge", "vanilla"):
            raise ValueError(f"disc_loss must be one of ['hinge', 'vanilla'], got {disc_loss}")
        self.discr_loss = get_discr_loss_fn(self.disc_loss_name)
        self.learn_logvar = False
        self.resize_input_to_target, self.resize_target_to_input = resize_input, resize_target
        self.forward_keys = ["global_step", "optimizer_idx", "split"]
        self.extra_log_keys = set(extra_log_keys or [])

    def get_trainable_parameters(self) -> Iterator[nn.Parameter]:
        yield from self.discr.parameters()

    def _discriminator_module(self) -> nn.Module:
        return self.discr
