"""AutoencoderKL holder: the slice of neurosis.models.autoencoder (autoencoder.py:429-522) the diffusion engine
touches -- construction, weight layout (encoder.*, decoder.*, quant_conv.*, post_quant_conv.*), freeze()/eval(),
encode() and decode().  The training step only runs the ENCODER forward (DiffusionEngine.encode_first_stage,
models/diffusion.py:186-197); the decoder forward serves sampling / log_images.  The VAE's own
training step (row N2) is not built.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ..modules.diffusion.model import Decoder, Encoder
from ..nn import Conv2d


class AutoencoderKL(nn.Module):
    def __init__(self, *, embed_dim: int, ddconfig: dict, loss=None, ckpt_path: Optional[str] = None, monitor: Optional[str] = None, **kwargs):
        super().__init__()
        dd = dict(ddconfig)
        dd.pop("standalone", None)
        self.embed_dim = embed_dim
        self.encoder = Encoder(**dd, embed_dim=embed_dim, standalone=False)
        self.decoder = Decoder(**dd, embed_dim=embed_dim, standalone=False)
        z = dd["z_channels"]
        double_z = dd.get("double_z", True)
        self.quant_conv = Conv2d((1 + double_z) * z, (1 + double_z) * embed_dim, 1)
        self.post_quant_conv = Conv2d(embed_dim, z, 1)
        self.monitor = monitor
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def init_from_ckpt(self, path: str) -> None:
        from safetensors.torch import load_file

        sd = load_file(path) if str(path).endswith(".safetensors") else torch.load(path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        own = self.state_dict()
        self.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)

    def freeze(self) -> None:
        for p in self.parameters():
            p.requires_grad = False

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """moments -> DiagonalGaussian mode (mean), fp32 NCHW."""
        enc = self.encoder
        prev = (enc.standalone, enc.quant_conv)
        enc.standalone, enc.quant_conv = True, self.quant_conv
        try:
            return enc(x, regularize=True)
        finally:
            enc.standalone, enc.quant_conv = prev

    @torch.no_grad()
    def decode(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        """post_quant_conv + decoder (autoencoder.py:505-508), fp32 NCHW."""
        dec = self.decoder
        prev = (dec.standalone, dec.post_quant_conv)
        dec.standalone, dec.post_quant_conv = True, self.post_quant_conv
        try:
            return dec(z, **kwargs)
        finally:
            dec.standalone, dec.post_quant_conv = prev
