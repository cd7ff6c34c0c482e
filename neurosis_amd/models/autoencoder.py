"""AutoencoderKL holder: the slice of neurosis.models.autoencoder (autoencoder.py:429-522) the SDXL training
step touches -- construction, weight layout (encoder.*, quant_conv.*), freeze()/eval().  The hot path only
runs the ENCODER forward (DiffusionEngine.encode_first_stage, models/diffusion.py:186-197); the decoder and
the VAE's own training step are outside SURVEY section 8(a) (row N2 of 8(f)) and are not built.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from ..modules.diffusion.model import Encoder
from ..nn import Conv2d


class AutoencoderKL(nn.Module):
    def __init__(self, *, embed_dim: int, ddconfig: dict, loss=None, ckpt_path: Optional[str] = None, monitor: Optional[str] = None, **kwargs):
        super().__init__()
        dd = dict(ddconfig)
        dd.pop("standalone", None)
        self.embed_dim = embed_dim
        self.encoder = Encoder(**dd, embed_dim=embed_dim, standalone=False)
        self.decoder = None  # out of scope (inference / VAE training only)
        z = dd["z_channels"]
        double_z = dd.get("double_z", True)
        self.quant_conv = Conv2d((1 + double_z) * z, (1 + double_z) * embed_dim, 1)
        self.post_quant_conv = None
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
