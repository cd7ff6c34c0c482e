"""CPU oracle for LPIPS over a VGG16 trunk -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Functional fp32 restatement of `LPIPS.forward` (modules/losses/perceptual.py:170-195; paths relative to
/root/reference/src/neurosis/) with lpips=True, spatial=False: ScalingLayer (:197-207), the VGG16 feature taps the reference
takes from torchvision (features.3/8/15/22/29 = relu1_2 ... relu5_3; torchvision is NOT installed here, the architecture is its
published cfg "D"), normalize_tensor (:215-217), squared difference, NetLinLayer 1x1 convolution (:198-212), spatial_average
(:220-221), summed over the layers.  Pinned by tests/golden/lpips_vgg_tiny.pt: the reference's own LPIPS.forward with its packaged
calibrated lin weights over a torch stand-in of the trunk with synthetic weights (the ImageNet weights are a download)."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
TAPS = (3, 8, 15, 22, 29)
SHIFT = torch.tensor([-0.030, -0.088, -0.188])[None, :, None, None]
SCALE = torch.tensor([0.458, 0.448, 0.450])[None, :, None, None]


def vgg_taps(sd: dict, x: Tensor) -> list:
    feats, i = [], 0
    for item in CFG:
        if item == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
        else:
            x = F.relu(F.conv2d(x, sd[f"pnet.features.{i}.weight"], sd[f"pnet.features.{i}.bias"], padding=1))
            i += 2
            if i - 1 in TAPS:
                feats.append(x)
    return feats


def lpips(sd: dict, lin: dict, x: Tensor, y: Tensor) -> Tensor:
    """[B, 1, 1, 1]"""
    fx, fy = vgg_taps(sd, (x - SHIFT) / SCALE), vgg_taps(sd, (y - SHIFT) / SCALE)
    total = 0
    for k, (a, b) in enumerate(zip(fx, fy)):
        a = a / (a.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
        b = b / (b.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
        total = total + F.conv2d((a - b).pow(2), lin[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return total
