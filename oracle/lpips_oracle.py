"""CPU oracle for LPIPS over the AlexNet / VGG16 trunks -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Functional fp32 restatement of `LPIPS.forward` (modules/losses/perceptual.py:170-195; paths relative to
/root/reference/src/neurosis/) with lpips=True, spatial=False: ScalingLayer (:197-207), the VGG16 feature taps the reference
takes from torchvision (features.3/8/15/22/29 = relu1_2 ... relu5_3; torchvision is NOT installed here, the architecture is its
published cfg "D"), normalize_tensor (:215-217), squared difference, NetLinLayer 1x1 convolution (:198-212), spatial_average
(:220-221), summed over the layers.  The AlexNet trunk (the reference's default pnet_type) taps features.1/4/7/9/11 of torchvision's
alexnet: conv 11x11/4 pad 2, ReLU, max-pool 3x3/2, conv 5x5 pad 2, ReLU, max-pool 3x3/2, three 3x3 convolutions with ReLUs.
Pinned by tests/golden/lpips_{vgg,alex}_tiny.pt: the reference's own LPIPS.forward with its packaged calibrated lin weights over a
torch stand-in of the trunk with synthetic weights (the ImageNet weights are a download)."""
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


def alex_taps(sd: dict, x: Tensor) -> list:
    def conv(i, h, stride, pad):
        return F.relu(F.conv2d(h, sd[f"pnet.features.{i}.weight"], sd[f"pnet.features.{i}.bias"], stride=stride, padding=pad))

    r1 = conv(0, x, 4, 2)
    r2 = conv(3, F.max_pool2d(r1, 3, 2), 1, 2)
    r3 = conv(6, F.max_pool2d(r2, 3, 2), 1, 1)
    r4 = conv(8, r3, 1, 1)
    r5 = conv(10, r4, 1, 1)
    return [r1, r2, r3, r4, r5]


def lpips(sd: dict, lin: dict, x: Tensor, y: Tensor, trunk: str = "vgg") -> Tensor:
    """[B, 1, 1, 1]"""
    taps = alex_taps if trunk == "alex" else vgg_taps
    fx, fy = taps(sd, (x - SHIFT) / SCALE), taps(sd, (y - SHIFT) / SCALE)
    total = 0
    for k, (a, b) in enumerate(zip(fx, fy)):
        a = a / (a.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
        b = b / (b.pow(2).sum(dim=1, keepdim=True).sqrt() + 1e-10)
        total = total + F.conv2d((a - b).pow(2), lin[f"lin{k}.model.1.weight"]).mean([2, 3], keepdim=True)
    return total
