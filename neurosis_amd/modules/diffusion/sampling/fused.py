"""The denoiser the engine hands to a sampler on MI355X.

Called like the reference's `denoiser_cb(inputs, sigma, c)` closure (models/diffusion.py:306-310) it is exactly that.  A
sampler from this package additionally finds `guided()` / `euler()`: the same evaluation with the elementwise work around the
network moved into the nk_sample_* kernels (csrc/sampling.hip):

    reference, per step (CFG):  cat([x]*2), cat([s]*2), cat(uc, c) | x*c_in | UNet | F*c_out + x*c_skip | chunk, u + s(c-u) |
                                (x - D)/sigma | x + dt*d                               ~12 launches over the latents, 2B-sized
    here:                       nk_sample_prepare | UNet (bf16 tokens in, bf16 tokens out) | nk_sample_euler_step

The latents stay fp32 NCHW at the API; the network side never leaves channels-last bf16 tokens, so the NCHW<->NHWC
transposes of the generic path disappear as well.

`euler()` can also capture the whole step -- prepare, UNet, guided Euler update -- into one hipGraph per (shape, guider,
conditioning shape, parameter epoch) and replay it: the step's only inputs are device tensors (x, sigma_hat[B], sigma_next[B],
conditioning), so nothing is baked into the graph but addresses (the stream-K GEMM keeps no per-launch state in its kernel
arguments for exactly this reason).  Opt-in (`use_graph=True` or NK_SAMPLE_GRAPH=1): measured on MI355X / ROCm 7.2, SDXL
1024^2 with CFG, the step is GPU-bound even at batch 1 (37 ms for a UNet batch of 2; the ~1 300 launches cost the host 37 ms
eager and 26 ms as one hipGraphLaunch), so replaying buys host time, not images per second, and pins one forward's
activations (25 GB at batch 1) for the life of the graph.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor

from .... import ops
from ....graphs import frozen_stamp
from ....lib import call
from ....nn import as_tokens
from ....ops import BF16, Img
from ...guidance import Guider, IdentityGuider, VanillaCFG
from ..denoiser import Denoiser
from ..wrappers import OpenAIWrapper


__all__ = ["CapturedEulerStep", "FusedDenoiser"]


class CapturedEulerStep:
    """One fused Euler step as a hipGraph over static buffers (latents, the two sigma vectors, stacked conditioning)."""

    def __init__(self, fused: "FusedDenoiser", x: Tensor, cond: dict, uc: dict, guider: Guider):
        B = x.shape[0]
        self.x = x.clone()
        self.sigma_hat = torch.ones(B, dtype=torch.float32, device=x.device)
        self.sigma_next = torch.ones(B, dtype=torch.float32, device=x.device)
        self.cond = {k: v.clone() for k, v in cond.items() if torch.is_tensor(v)}
        self.uc = {k: uc[k].clone() for k in self.cond} if guider.rep == 2 else self.cond
        stream = torch.cuda.Stream(device=x.device)
        stream.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(stream), ops.capture_scope():
            # eager run on the capture stream first: weight shadows, padded-conv parameters and the stream-K workspace of this
            # stream are created here (allocations that must not happen while capturing)
            fused._euler_eager(self.x, self.sigma_hat, self.sigma_next, self.cond, self.uc, guider, out=torch.empty_like(self.x))
        torch.cuda.current_stream(x.device).wait_stream(stream)
        torch.cuda.synchronize(x.device)
        self.graph = torch.cuda.CUDAGraph()
        # ops.capture_scope: weight-derived caches (the channel-padded stand-ins of the 4-channel convolutions and their bf16
        # shadows) are refilled IN PLACE inside the graph, so a replay after an optimizer step or inside ema_scope() reads the
        # current masters -- the key of this graph (frozen_stamp) does not change for store-managed parameters
        with ops.capture_scope(), torch.cuda.graph(self.graph, stream=stream):
            fused._euler_eager(self.x, self.sigma_hat, self.sigma_next, self.cond, self.uc, guider, out=self.x)

    def load_conditioning(self, cond: dict, uc: dict) -> None:
        for k, buf in self.cond.items():
            buf.copy_(cond[k])
        if self.uc is not self.cond:
            for k, buf in self.uc.items():
                buf.copy_(uc[k])

    def __call__(self, x: Tensor, sigma_hat: Tensor, sigma_next: Tensor) -> Tensor:
        if x.data_ptr() != self.x.data_ptr():      # (after the first step the sampler hands our own buffer back)
            self.x.copy_(x)
        self.sigma_hat.copy_(sigma_hat)
        self.sigma_next.copy_(sigma_next)
        self.graph.replay()
        return self.x


class FusedDenoiser:
    def __init__(self, network, denoiser: Denoiser, use_graph: bool | None = None, **model_kwargs):
        self.network, self.denoiser, self.model_kwargs = network, denoiser, model_kwargs
        self.use_graph = os.environ.get("NK_SAMPLE_GRAPH", "0") == "1" if use_graph is None else use_graph
        self._captured: dict = {}
        self._loaded = None

    # the reference's callback ---------------------------------------------------------------------
    def __call__(self, inputs: Tensor, sigma: Tensor, c: dict) -> Tensor:
        return self.denoiser(self.network, inputs, sigma, c, "D", **self.model_kwargs)

    # the fused route ------------------------------------------------------------------------------
    def supports(self, x: Tensor, guider: Guider, cond: dict) -> bool:
        if type(guider) not in (VanillaCFG, IdentityGuider) or not isinstance(self.network, OpenAIWrapper):
            return False
        return x.dim() == 4 and x.dtype == torch.float32 and self.network.fused_unet(x, cond, self.model_kwargs) is not None

    def _network(self, x: Tensor, sigma: Tensor, cond: dict, uc: dict, guider: Guider):
        """one UNet evaluation on rep stacked copies of c_in * x -> (bf16 output tokens Img, c_skip[B], c_out[B])"""
        B, C, H, W = x.shape
        rep = guider.rep
        unet = self.network.diffusion_model
        c_skip, c_out, c_in, c_noise = (t.contiguous() for t in self.denoiser.coefficients(sigma))
        c_skip, c_out, c_in = c_skip.float(), c_out.float(), c_in.float()
        cpad = (C + 7) // 8 * 8
        net_in = torch.empty(rep * B * H * W, cpad, dtype=BF16, device=x.device)
        call("nk_sample_prepare", x.data_ptr(), c_in.data_ptr(), net_in.data_ptr(), B, C, H * W, cpad, rep, ops._stream())

        def stacked(key):
            value = cond.get(key)
            if value is None:
                return None
            return as_tokens(value if rep == 1 else torch.cat((uc[key], value), 0))

        timesteps = c_noise if rep == 1 else torch.cat((c_noise, c_noise))
        out, _ = unet.fwd(Img(net_in, rep * B, H, W), timesteps, stacked("crossattn"), stacked("vector"))
        return out, c_skip, c_out

    def guided(self, x: Tensor, sigma: Tensor, cond: dict, uc: dict, guider: Guider) -> Tensor:
        """guider(denoiser(*guider.prepare_inputs(x, sigma, cond, uc)), sigma) as fp32 NCHW"""
        x = x.contiguous()
        B, C, H, W = x.shape
        out, c_skip, c_out = self._network(x, sigma, cond, uc, guider)
        denoised = torch.empty_like(x)
        call("nk_sample_denoise", out.t.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), float(getattr(guider, "scale", 1.0)),
             denoised.data_ptr(), B, C, H * W, out.C, guider.rep, ops._stream())
        return denoised

    def _euler_eager(self, x: Tensor, sigma_hat: Tensor, next_sigma: Tensor, cond: dict, uc: dict, guider: Guider, out: Tensor) -> Tensor:
        B, C, H, W = x.shape
        net, c_skip, c_out = self._network(x, sigma_hat, cond, uc, guider)
        sh, sn = sigma_hat.float().contiguous(), next_sigma.float().contiguous()
        call("nk_sample_euler_step", net.t.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), sh.data_ptr(), sn.data_ptr(),
             float(getattr(guider, "scale", 1.0)), out.data_ptr(), None, B, C, H * W, net.C, guider.rep, ops._stream())
        return out

    def euler(self, x: Tensor, sigma_hat: Tensor, next_sigma: Tensor, cond: dict, uc: dict, guider: Guider) -> Tensor:
        """x + (next_sigma - sigma_hat) * (x - D) / sigma_hat with D as in `guided`.  With graphs on, the returned tensor is the
        captured step's own latent buffer (overwritten by the next call)."""
        x = x.contiguous()
        if not self.use_graph:
            return self._euler_eager(x, sigma_hat, next_sigma, cond, uc, guider, out=torch.empty_like(x))
        tensors = {k: v for k, v in cond.items() if torch.is_tensor(v)}
        # the graph holds ADDRESSES, of the weights' bf16 shadows too: a flat store rewrites its shadows in place (its identity is
        # the key), free parameters get a new shadow buffer whenever their version or address changes (graphs.frozen_stamp)
        key = (tuple(x.shape), x.device, type(guider), float(getattr(guider, "scale", 1.0)), frozen_stamp(self.network.diffusion_model),
               tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tensors.items())))
        step = self._captured.get(key)
        if step is None:
            self._captured.clear()                 # one live graph: each pins a forward's worth of activations
            step = self._captured[key] = CapturedEulerStep(self, x, cond, uc, guider)
            self._loaded = None
        # conditioning is copied into the graph's buffers once per (cond, uc) pair, not per step
        ident = (key, tuple(v.data_ptr() for v in tensors.values()), tuple(uc[k].data_ptr() for k in tensors), tuple(v._version for v in tensors.values()),
                 tuple(uc[k]._version for k in tensors))
        if self._loaded != ident:
            step.load_conditioning(cond, uc)
            self._loaded = ident
        return step(x, sigma_hat, next_sigma)
