"""PatchGAN discriminator (`neurosis.modules.losses.patchgan.model`, :6-95; Pix2Pix's NLayerDiscriminator): 4x4 convolutions,
stride 2 down to the last-but-one, BatchNorm2d + LeakyReLU(0.2) between them, a 1-channel logit map out.  Same module list and
state_dict keys (`layers.N.weight`, BatchNorm `running_mean` / `running_var` / `num_batches_tracked`).

On MI355X: channels-last bf16 tokens end to end; the convolutions are the implicit-GEMM tile engine (4x4 taps, 3 -> 64 input and
512 -> 1 output channels zero-padded to multiples of 8), each BatchNorm runs in training mode with its LeakyReLU fused
(csrc/gan.hip: column statistics -> apply, three HBM passes forward, four backward), and `fwdb` keeps the backward closures:
weight gradients for the discriminator's own update, the input gradient for the generator's adversarial term.
ActNorm (`use_actnorm=True`) is not built.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from ... import ops
from ...nn import Conv2d
from ...ops import Img

SLOPE = 0.2


def weights_init(m: nn.Module) -> None:
    """N(0, 0.02) convolution weights, zero biases; BatchNorm weights N(1, 0.02) (reference :6-18)"""
    if isinstance(m, (Conv2d, nn.modules.conv._ConvNd, nn.Linear)) and hasattr(m, "weight"):
        nn.init.normal_(m.weight.data, 0.0, 0.02)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias.data, 0.0)
    elif isinstance(m, nn.modules.batchnorm._BatchNorm):
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0.0)


def _no_backward(_):
    raise RuntimeError("the discriminator was run in eval mode (running BatchNorm statistics): no backward through it")


class NLayerDiscriminator(nn.Module):
    def __init__(self, input_nc: int = 3, ndf: int = 64, n_layers: int = 3, use_actnorm: bool = False):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("ActNorm discriminators are not built (the autoencoder configs use BatchNorm)")
        self.input_nc = input_nc
        layers = [Conv2d(input_nc, ndf, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(SLOPE, True)]
        width = ndf
        for n in range(1, n_layers + 1):
            wider = ndf * min(2**n, 8)
            layers += [Conv2d(width, wider, kernel_size=4, stride=2 if n < n_layers else 1, padding=1, bias=False), nn.BatchNorm2d(wider),
                       nn.LeakyReLU(SLOPE, True)]
            width = wider
        layers.append(Conv2d(width, 1, kernel_size=4, stride=1, padding=1))
        self.layers = nn.ModuleList(layers)

    def initialize_weights(self):
        return self.apply(weights_init)

    def fwdb(self, x: Img, need_dx: bool = True):
        """x: image tokens (channels padded to 8).  Returns (logits Img with 8 channels, the first is real; bwd) with
        bwd(d_logits tokens [M, 8]) -> d_image tokens or None."""
        tape = []
        h = x
        i, first = 0, True
        while i < len(self.layers):
            conv = self.layers[i]
            h, b_conv = conv.fwd(h, need_dx=need_dx or not first)
            first = False
            tape.append(lambda d, b=b_conv: (lambda r: None if r is None else r.t)(b(d)[0]))
            nxt = self.layers[i + 1] if i + 1 < len(self.layers) else None
            if isinstance(nxt, nn.BatchNorm2d):
                if self.training:
                    nxt.num_batches_tracked += 1
                    t, b_bn = ops.batchnorm_fwd(h.t, nxt.weight, nxt.bias, nxt.running_mean, nxt.running_var, nxt.eps, nxt.momentum, SLOPE)
                else:
                    # evaluation (logging): the running statistics as a per-channel affine map, one HIP launch; no gradients (nothing
                    # trains through an eval-mode discriminator)
                    t = ops.batchnorm_eval(h.t, nxt.weight.detach(), nxt.bias.detach(), nxt.running_mean, nxt.running_var, nxt.eps, SLOPE)
                    b_bn = _no_backward
                h = Img(t, h.N, h.H, h.W)
                tape.append(b_bn)
                i += 3
            elif isinstance(nxt, nn.LeakyReLU):
                t, b_act = ops.leaky_relu_fwd(h.t, SLOPE)
                h = Img(t, h.N, h.H, h.W)
                tape.append(b_act)
                i += 2
            else:
                i += 1

        def bwd(d_logits: Tensor):
            d = d_logits
            for b in reversed(tape):
                d = b(d)
            tape.clear()
            return d

        return h, bwd

    def forward(self, x: Tensor) -> Tensor:
        """NCHW image -> NCHW logits [B, 1, h, w] (fp32), batch statistics in train mode, running statistics in eval mode.  This entry
        point is for evaluation / logging (the reference's loss.log_images); training goes through fwdb (AutoencodingEngine)."""
        with torch.no_grad():
            B, C, H, W = x.shape
            logits, _ = self.fwdb(Img(ops.nchw_to_tokens(x.float().contiguous(), (C + 7) // 8 * 8), B, H, W), need_dx=False)
            return ops.tokens_to_nchw(logits.t, B, 1, logits.H, logits.W, dtype=torch.float32)
