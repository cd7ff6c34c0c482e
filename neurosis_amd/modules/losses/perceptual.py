"""LPIPS perceptual distance (`neurosis.modules.losses.perceptual`, :64-228; Zhang et al. 2018, v0.1) over the AlexNet (the
reference's default) or the VGG16 trunk.

    d(x, y) = sum_layers mean_pixels sum_c w_c (unit(f_x)_c - unit(f_y)_c)^2,   f = the trunk's five tapped ReLU outputs

The reference takes the trunk from torchvision (`create_alexnet_extractor` / `create_vgg_extractor`, extractors.py:11-30) and the
calibrated `lin` weights from its package data; here the trunk is the same convolutions under torchvision's parameter names
(`pnet.features.N.*`) on the implicit-GEMM tile engine (AlexNet's 11x11 / stride 4 and 5x5 convolutions included), ReLU and max-pool
(2x2 / 2 for VGG, overlapping 3x3 / 2 for AlexNet) as HIP kernels, and each layer's normalise / difference / 1x1 lin / spatial mean
is ONE kernel (`nk_lpips_layer_fwd`) instead of six feature-map-sized passes.  `fwdb` also returns the backward w.r.t. the SECOND
image (the reconstruction): the trunk is frozen, so its convolutions only pass gradients through (no weight gradients).

Weights: there is no hub access here.  `lin_weights` takes a state dict / .safetensors path with the `linN.model.1.weight`
tensors (the reference ships them as neurosis/data/lpips/{alex,vgg}_lpips_v0.1.safetensors; NK_LPIPS_WEIGHTS may name that file
or its directory -- the reference package itself is never imported); the trunk's ImageNet weights load through `load_state_dict` (torchvision's alexnet / vgg16 keys) --
without them the trunk is random, which `pnet_rand=True` makes explicit.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch import Tensor, nn

from ... import ops
from ...nn import Conv2d
from ...ops import BF16, Img

# torchvision layer lists by position in `.features`: ("conv", out channels, kernel, stride, padding) | "relu" | ("pool", kernel, stride)
VGG16_PLAN = tuple(item for c in (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
                   for item in ((("pool", 2, 2),) if c == "M" else (("conv", c, 3, 1, 1), "relu")))[:30]      # cfg "D" up to relu5_3
ALEX_PLAN = (("conv", 64, 11, 4, 2), "relu", ("pool", 3, 2), ("conv", 192, 5, 1, 2), "relu", ("pool", 3, 2),
             ("conv", 384, 3, 1, 1), "relu", ("conv", 256, 3, 1, 1), "relu", ("conv", 256, 3, 1, 1), "relu")   # alexnet().features[:12]
# the reference's PNET_CONFIG (perceptual.py:35-60): channels of the five taps and the features.N each is taken at
PNET_CONFIG = {
    "alex": {"channels": (64, 192, 384, 256, 256), "taps": {1: "relu1", 4: "relu2", 7: "relu3", 9: "relu4", 11: "relu5"}, "plan": ALEX_PLAN},
    "vgg": {"channels": (64, 128, 256, 512, 512), "taps": {3: "relu1", 8: "relu2", 15: "relu3", 22: "relu4", 29: "relu5"}, "plan": VGG16_PLAN},
}
VGG_TAPS, VGG_CHANNELS = PNET_CONFIG["vgg"]["taps"], PNET_CONFIG["vgg"]["channels"]


class _Trunk(nn.Module):
    """torchvision's alexnet().features / vgg16().features by position: Conv2d at the convolution slots (so the state-dict keys are
    torchvision's `features.N.weight / bias`), placeholders at the ReLU and pooling slots."""

    def __init__(self, plan, taps):
        super().__init__()
        layers, cin = [], 3
        for item in plan:
            if item != "relu" and item[0] == "conv":
                _, cout, k, stride, pad = item
                layers.append(Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad))
                cin = cout
            else:
                layers.append(nn.Identity())                      # ReLU / MaxPool2d
        self.features = nn.ModuleList(layers)
        self.plan, self.taps = tuple(plan), dict(taps)

    def run(self, x: Img, keep_tape: bool):
        """(taps: {name: Img}, tape: [(conv bwd | None, relu / pool bwd, tap name or None)])"""
        taps, tape, h, i = {}, [], x, 0
        while i < len(self.plan):
            item = self.plan[i]
            if item != "relu" and item[0] == "conv":              # a convolution and the ReLU behind it
                h, b_conv = self.features[i].fwd(h, need_dx=keep_tape)
                act, b_act = ops.leaky_relu_fwd(h.t, 0.0)
                h = Img(act, h.N, h.H, h.W)
                name = self.taps.get(i + 1)
                if name is not None:
                    taps[name] = h
                if keep_tape:
                    tape.append((b_conv, b_act, name))
                i += 2
            else:
                h, b_pool = ops.maxpool_fwd(h, item[1], item[2])
                if keep_tape:
                    tape.append((None, b_pool, None))
                i += 1
        return taps, tape


class NetLinLayer(nn.Module):
    """a 1x1 convolution to one channel without bias (reference :198-212); only its weight vector is used here"""

    def __init__(self, chn_in: int, chn_out: int = 1, use_dropout: bool = False):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout() if use_dropout else nn.Identity(), nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False))


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([0.458, 0.448, 0.450])[None, :, None, None])

    def forward(self, inp: Tensor) -> Tensor:
        return (inp - self.shift) / self.scale


class LPIPS(nn.Module):
    def __init__(self, pnet_type: str = "alex", pretrained: bool = True, lpips: bool = True, pnet_rand: bool = False, pnet_tune: bool = False,
                 use_dropout: bool = False, spatial: bool = False, freeze: bool = True, verbose: bool = False, lin_weights=None):
        super().__init__()
        if "vgg" in pnet_type:
            pnet_type = "vgg"
        if "alex" in pnet_type:
            pnet_type = "alex"
        if pnet_type not in PNET_CONFIG:
            raise KeyError(pnet_type)
        if pnet_tune or spatial or not lpips:
            raise NotImplementedError("pnet_tune / spatial / lpips=False are not used by the autoencoder loss and are not built")
        conf = PNET_CONFIG[pnet_type]
        self.pnet_type, self.pnet_tune, self.pnet_rand, self.lpips, self.spatial = pnet_type, pnet_tune, pnet_rand, lpips, spatial
        self.scaling_layer = ScalingLayer()
        self.chns, self.L = list(conf["channels"]), len(conf["channels"])
        self.pnet_keys = list(conf["taps"].values())
        self.pnet = _Trunk(conf["plan"], conf["taps"])
        self.lin0, self.lin1, self.lin2, self.lin3, self.lin4 = (NetLinLayer(c, use_dropout=use_dropout) for c in self.chns)
        self.lins = nn.ModuleDict(dict(zip(self.pnet_keys, (self.lin0, self.lin1, self.lin2, self.lin3, self.lin4))))
        if pretrained:
            self._load_pretrained(lin_weights)
        if freeze:
            self.requires_grad_(False)

    def _load_pretrained(self, source) -> None:
        if source is None:
            source = os.environ.get("NK_LPIPS_WEIGHTS")             # a directory holding {alex,vgg}_lpips_v0.1.safetensors, or one such file
            if source and os.path.isdir(source):
                source = os.path.join(source, f"{self.pnet_type}_lpips_v0.1.safetensors")
            if not source or not os.path.exists(source):
                # (the product path never imports the reference package: the weights are named, not looked up through it)
                raise RuntimeError("LPIPS(pretrained=True) needs the calibrated lin weights: pass lin_weights= (a state dict or the path of "
                                   "{alex,vgg}_lpips_v0.1.safetensors; the reference ships them under neurosis/data/lpips/), set NK_LPIPS_WEIGHTS "
                                   "to that file or its directory, or use pretrained=False")
        if not isinstance(source, dict):
            from safetensors.torch import load_file

            source = load_file(str(source))
        self.load_state_dict({k: v for k, v in source.items() if k.startswith("lin")}, strict=False)

    def _lin_vectors(self):
        return [getattr(self, f"lin{i}").model[1].weight.detach().float().reshape(-1).contiguous() for i in range(self.L)]

    def _scaled_tokens(self, img: Tensor) -> Img:
        """fp32 NCHW image in [-1, 1] -> ScalingLayer -> channels-last bf16 tokens (3 channels padded to 8)"""
        B, C, H, W = img.shape
        return Img(ops.nchw_to_tokens(self.scaling_layer(img.float()).contiguous(), 8), B, H, W)

    def fwdb(self, x: Tensor, y_tokens: Img):
        """x: fp32 NCHW reference image; y_tokens: the other image as UNSCALED bf16 tokens [B*H*W, 8] (e.g. the decoder's output).
        Returns (distance [B] fp32, bwd) with bwd(upstream [B] fp32) -> d sum_b upstream_b * distance_b / d y_tokens."""
        B = x.shape[0]
        dev = x.device
        shift8 = torch.zeros(8, device=dev)
        scale8 = torch.ones(8, device=dev)
        shift8[:3], scale8[:3] = self.scaling_layer.shift.reshape(-1), self.scaling_layer.scale.reshape(-1)
        y_scaled = ops.cast_bf16(((y_tokens.t.float() - shift8) / scale8).contiguous())      # (padding channels: (0 - 0) / 1)
        taps_x, _ = self.pnet.run(self._scaled_tokens(x), keep_tape=False)
        taps_y, tape = self.pnet.run(Img(y_scaled, y_tokens.N, y_tokens.H, y_tokens.W), keep_tape=True)
        out = torch.empty(B, dtype=torch.float32, device=dev)
        layer_bwd = {}
        for i, (name, w) in enumerate(zip(self.pnet_keys, self._lin_vectors())):
            layer_bwd[name] = ops.lpips_layer(taps_x[name], taps_y[name], w, out, accumulate=i > 0)

        def bwd(upstream: Tensor) -> Tensor:
            up = upstream.float().contiguous()
            d = None
            for b_conv, b_other, name in reversed(tape):
                if b_conv is None:                         # pool
                    d = b_other(d)
                    continue
                if name is not None:                       # a tapped activation: the layer distance's own gradient joins in
                    g = layer_bwd[name](up)
                    d = g if d is None else ops.add(d, g)
                d = b_conv(b_other(d))[0].t
            tape.clear()
            return ops.cast_bf16((d.float() / scale8).contiguous())

        return out, bwd

    @torch.no_grad()
    def forward(self, x: Tensor, y: Tensor, retPerLayer: bool = False, normalize: bool = False) -> Tensor:
        """[B, 1, 1, 1] fp32, as the reference's default (lpips=True, spatial=False) path"""
        if retPerLayer:
            raise NotImplementedError("retPerLayer is not built")
        if normalize:
            x, y = x.mul(2.0).add(-1.0), y.mul(2.0).add(-1.0)
        B, C, H, W = y.shape
        out, _ = self.fwdb(x, Img(ops.nchw_to_tokens(y.float().contiguous(), 8), B, H, W))
        return out.reshape(B, 1, 1, 1)
