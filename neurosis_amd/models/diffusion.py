"""DiffusionEngine: the training-step and sampling slices of neurosis.models.diffusion.DiffusionEngine
(/root/reference/src/neurosis/models/diffusion.py:35-233 and :172-184,298-420) without Lightning.

What is mirrored: the constructor's model wiring (OpenAIWrapper around the UNet, the VAE dismantled into
`vae_encoder` with its quant_conv, models/diffusion.py:73,146-164), get_input, encode_first_stage, forward and
training_step, and the `model.diffusion_model.*` / `vae_encoder.*` state_dict prefixes.  What replaces
Lightning: `training_step` returns loss.mean() exactly as the reference does; `optimizer_step` and the
data-parallel exchange are explicit methods because there is no Trainer (bench.py and the tests drive them).
Sampling: `sample`, `decode_first_stage`, `ema_scope` and `log_images` (tensors only: the text-as-image
panels of `log_conditionings` are logging, not compute).  Hooks, loggers and checkpoint IO are outside section 8.
"""
from __future__ import annotations

import os
from contextlib import contextmanager, nullcontext
from math import ceil
from typing import Callable, Optional

import torch
from torch import Tensor, nn

from .. import ops
from ..modules.diffusion import Denoiser, DiffusionLoss, OpenAIWrapper, UNetModel
from ..nn import FlatParamStore
from .autoencoder import AutoencoderKL


class PrecomputedConditioner(nn.Module):
    """Stand-in for GeneralConditioner (modules/encoders/embedding.py:90-149, frozen text encoders: out of scope):
    returns the conditioning tensors the batch already carries."""

    def forward(self, batch: dict, force_zero_embeddings=None) -> dict:
        zeroed = set(force_zero_embeddings or ())
        out = {}
        for k in ("crossattn", "vector", "concat"):
            if k in batch and batch[k] is not None:
                out[k] = torch.zeros_like(batch[k]) if k in zeroed else batch[k]
        return out


class DiffusionEngine(nn.Module):
    def __init__(self, model: UNetModel, denoiser: Denoiser, first_stage_model: Optional[AutoencoderKL], conditioner: Optional[nn.Module] = None,
                 sampler=None, optimizer=None, scheduler=None, loss_fn: Optional[DiffusionLoss] = None, ckpt_path=None, use_ema: bool = False,
                 scale_factor: float = 1.0, disable_first_stage_autocast: bool = False, input_key: str = "jpg", vae_batch_size: Optional[int] = None,
                 log_sigmas: bool = False, **kwargs):
        super().__init__()
        self.use_ema = use_ema                                    # models/diffusion.py:93-99; created in setup_flat_params()
        self.ema_decay_rate = kwargs.pop("ema_decay_rate", 0.9999)
        self.input_key = input_key
        self.model = OpenAIWrapper(model)
        self.denoiser = denoiser
        self.conditioner = conditioner if conditioner is not None else PrecomputedConditioner()
        self.loss_fn = loss_fn
        self.scale_factor = scale_factor
        self.vae_batch_size = vae_batch_size
        self.log_sigmas = log_sigmas
        self.sampler = sampler
        # models/diffusion.py:43-44,78-79: LightningCLI hands over callables (params -> Optimizer, Optimizer -> LRScheduler);
        # configure_optimizers() below calls them as the reference's does (:261-296)
        self.optimizer = optimizer
        self.scheduler = scheduler
        self._torch_optimizer = None
        self._torch_scheduler = None
        self.vae_encoder = None
        self.vae_decoder = None
        if first_stage_model is not None:
            self._init_first_stage(first_stage_model)
        self.global_step = 0
        self._ckpt_path = ckpt_path
        # optimizer_step() runs on its own stream and is joined right before the next UNet forward (see optimizer_step)
        self.overlap_optimizer = os.environ.get("NK_OPT_OVERLAP", "1") != "0"
        self._optimizer_stream: Optional[torch.cuda.Stream] = None
        self._optimizer_in_flight = False
        # stream_optimizer = True: stream the update of each top-level UNet block behind that block's backward (_grads_ready).  Off by
        # default -- measured 187.4 vs 184.9 ms/step (interleaved): the HBM-bound update steals bandwidth from a backward whose two streams
        # already contend, while after backward it overlaps the next step's frozen VAE encoder for free.
        self.stream_optimizer = False
        self._streaming_step = False
        self._last_micro_batch = True
        self.store: Optional[FlatParamStore] = None
        self.last_log: dict = {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def init_from_ckpt(self, path) -> tuple:
        """models/diffusion.py:127-144: restore from a .safetensors file or a Lightning checkpoint (its "state_dict"), non-strict;
        `first_stage_model.*` keys of a full SDXL checkpoint are expected leftovers (the VAE lives under vae_encoder / vae_decoder
        here as in the reference).  Returns (missing, unexpected) after that filtering."""
        from pathlib import Path

        path = Path(path)
        if path.suffix == ".safetensors":
            from safetensors.torch import load_file

            sd = load_file(str(path))
        elif path.suffix in (".ckpt", ".pt", ".pth"):
            sd = torch.load(path, map_location="cpu", weights_only=False)
            sd = sd.get("state_dict", sd)
        else:
            raise NotImplementedError(f"Unknown checkpoint extension {path.suffix}")
        self.join_optimizer()
        missing, unexpected = self.load_state_dict(sd, strict=False)
        unexpected = [k for k in unexpected if not k.startswith("first_stage_model")]
        missing = [k for k in missing if not k.startswith("vae_") and "._orig_mod." not in k]
        if self.store is not None:
            self.store.refresh()
        return missing, unexpected

    def _init_first_stage(self, model: AutoencoderKL) -> None:
        """models/diffusion.py:146-164: keep the encoder, move quant_conv onto it (the reference needs
        ddconfig.standalone=true for this not to crash; here it always works)."""
        model = model.eval()
        model.freeze()
        enc = model.encoder
        enc.quant_conv = model.quant_conv
        enc.standalone = True
        self.vae_encoder = enc
        dec = getattr(model, "decoder", None)
        if dec is not None:
            dec.post_quant_conv = model.post_quant_conv
            dec.standalone = True
            self.vae_decoder = dec

    def setup_flat_params(self) -> FlatParamStore:
        """Re-home the trainable UNet parameters into the flat fp32/bf16/grad buffers (call after .cuda())."""
        self.store = FlatParamStore([p for p in self.model.diffusion_model.parameters() if p.requires_grad])
        # contract: every parameter gradient is OVERWRITTEN by its producer on the first micro-batch of a step
        # (store.state.grad_accumulate False) and added to on later ones, so the 10 GB buffer is never zero-filled between steps
        self.store.state.assume_zeroed = False
        if self.store.master.is_cuda:
            self.store.state.wgrad_stream = torch.cuda.Stream(device=self.store.master.device)
        if self.use_ema:
            self.configure_ema(self.ema_decay_rate)
        if self.optimizer is not None and self._torch_optimizer is None:
            self.configure_optimizers()
        self.stream_optimizer = self.stream_optimizer      # (re-)installs the UNet's gradient-ready hook if streaming is on
        return self.store

    @property
    def stream_optimizer(self) -> bool:
        """Apply the fused Adafactor block by block behind backward (`stream_optimizer = True`, `bench.py --stream-optimizer`; measured no faster, off by default).  It
        needs the UNet's gradient-ready hook, and a hook keeps the chain out of hipGraph replay (a FlatDataParallel wrapper
        takes the hook over when N > 1)."""
        return self._stream_optimizer

    @stream_optimizer.setter
    def stream_optimizer(self, on: bool) -> None:
        self._stream_optimizer = bool(on)
        unet = getattr(getattr(self, "model", None), "diffusion_model", None)
        if unet is None or getattr(self, "store", None) is None:
            return
        if on:
            if unet.grad_ready_hook is None:
                unet.grad_ready_hook = self._grads_ready
        elif unet.grad_ready_hook == self._grads_ready:
            unet.grad_ready_hook = None

    def _block_boundaries(self) -> list:
        """first tensor index (in store order) of every top-level UNet block: where the fused optimizer may cut its chunks"""
        unet = self.model.diffusion_model
        index = {id(p): i for i, p in enumerate(self.store.params)}
        tops = [unet.time_embed, getattr(unet, "label_emb", None), *unet.input_blocks, unet.middle_block, *unet.output_blocks, unet.out]
        out = []
        for m in tops:
            if m is None:
                continue
            idx = [index[id(p)] for p in m.parameters() if id(p) in index]
            if idx:
                out.append(min(idx))
        return sorted(set(out))

    def _grads_ready(self, module: nn.Module) -> None:
        """UNetModel.grad_ready_hook: `module`'s parameter gradients have been enqueued (main stream + weight-gradient stream).
        With the fused Adafactor, no data-parallel exchange and this being the step's last micro-batch, its slice of the update is
        issued NOW on the optimizer stream: the HBM-bound update (22 B per parameter) runs beside the MFMA-bound rest of backward
        instead of after it.  Same kernels on the same data as the one-shot update (per-tensor statistics never cross a block)."""
        af = getattr(self, "adafactor", None)
        if af is None or not self.stream_optimizer or not self._last_micro_batch or not self.store.master.is_cuda:
            return
        if self.model.diffusion_model.grad_ready_hook != self._grads_ready:       # a data-parallel wrapper owns the hook
            return
        idx = [p._nk_index for p in module.parameters() if getattr(p, "_nk_store", None) is self.store]
        if not idx:
            return
        chunks = af.chunks_in(min(idx), max(idx) + 1)
        if not chunks:
            return
        if self._optimizer_stream is None:
            self._optimizer_stream = torch.cuda.Stream(device=self.store.master.device)
        if not self._streaming_step:
            self.join_optimizer()
            af.begin_step()
            self._streaming_step = True
        opt = self._optimizer_stream
        opt.wait_stream(torch.cuda.current_stream())
        side = self.store.state.wgrad_stream
        if side is not None:
            opt.wait_stream(side)
        with torch.cuda.stream(opt):
            for ci in chunks:
                af.step_chunk(ci, 1.0)
        self._optimizer_in_flight = True

    def configure_optimizers(self):
        """models/diffusion.py:261-296: one parameter group for the UNet (plus `initial_lr` from `model.base_lr`), one per
        trainable embedder; `self.optimizer(param_groups)`, then `self.scheduler(optimizer)`; the same return value.  What
        comes back must be one of this package's fused optimizers (`neurosis_amd.optimizers.Adafactor` -- the class the example
        configs name under the prefix swap -- or `.AdamW`): the step is a few HIP launches over the flat buffers, and an eager
        torch optimizer walking 1 700 parameter views would silently replace it, so anything else is refused."""
        if self.optimizer is None:
            return None
        from ..optimizers import Adafactor, AdamW

        unet_params = {"name": "UNet", "params": [p for p in self.model.parameters() if p.requires_grad]}
        if getattr(self.model, "base_lr", None) is not None:
            unet_params["initial_lr"] = self.model.base_lr
        if self.store is not None and self.stream_optimizer:
            unet_params["chunk_boundaries"] = self._block_boundaries()     # lets the fused update be streamed block by block
        param_groups = [unet_params]
        for embedder in getattr(self.conditioner, "embedders", ()):
            if getattr(embedder, "is_trainable", False):
                raise NotImplementedError("trainable conditioner embedders are outside the fused training step (SURVEY.md section 8: frozen TE/VAE)")
        opt = self.optimizer(param_groups) if callable(self.optimizer) and not isinstance(self.optimizer, torch.optim.Optimizer) else self.optimizer
        if not isinstance(opt, (Adafactor, AdamW)):
            raise TypeError(f"DiffusionEngine: optimizer {type(opt).__module__}.{type(opt).__name__} cannot be fused; use "
                            "neurosis_amd.optimizers.Adafactor (the example configs' optimizer under the class_path prefix swap) or neurosis_amd.optimizers.AdamW")
        self._torch_optimizer = opt
        if self.store is not None and isinstance(opt, Adafactor):
            self.adafactor = opt.flat            # bound to the flat buffers now; optimizer_step() drives it
        if self.scheduler is not None:
            self._torch_scheduler = self.scheduler(opt) if callable(self.scheduler) and not hasattr(self.scheduler, "get_last_lr") else self.scheduler
            return {"optimizer": opt, "lr_scheduler": {"scheduler": self._torch_scheduler, "interval": "step"}}
        return opt

    def get_input(self, batch: dict) -> Tensor:
        inputs = batch[self.input_key]
        if inputs.ndim == 3:
            inputs = inputs.unsqueeze(0)
        return inputs

    @torch.no_grad()
    def encode_first_stage(self, x: Tensor) -> Tensor:
        """models/diffusion.py:186-197."""
        n_samples = self.vae_batch_size or x.shape[0]
        outs = [self.vae_encoder(x[n * n_samples:(n + 1) * n_samples], regularize=True) for n in range(ceil(x.shape[0] / n_samples))]
        z = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return self.scale_factor * z

    @torch.no_grad()
    def decode_first_stage(self, z: Tensor) -> Tensor:
        """models/diffusion.py:172-184: latents -> images (fp32 NCHW), in chunks of vae_batch_size."""
        z = 1.0 / self.scale_factor * z
        n_samples = self.vae_batch_size or z.shape[0]
        outs = [self.vae_decoder(z[n * n_samples:(n + 1) * n_samples], cat_zero=True) for n in range(ceil(z.shape[0] / n_samples))]
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def forward(self, x: Tensor, batch: dict, return_dict: bool = False, cond: Optional[dict] = None, **inject):
        if cond is None:
            cond = self.conditioner(batch)
        self.join_optimizer()          # everything above (VAE encode, conditioner) did not need the UNet's new weights
        return self.loss_fn._forward(self.model, self.denoiser, cond, x, batch, return_dict, **inject)

    def training_step(self, batch: dict, batch_idx: int = 0, **inject) -> Tensor:
        """models/diffusion.py:205-233.  `inject` may carry sigmas= / noise=.
        The frozen conditioner runs on a side stream BESIDE the frozen VAE encoder and is joined before the UNet (NK_COND_OVERLAP=0: in line,
        as the reference orders them): its towers are chains of small launches that leave most of the chip idle.  Steady-state step -1.0 ms
        (p50 152.9 vs 153.9, two alternating pairs on one box).  Round 3 had measured nothing for the same idea -- then the towers ran one
        after the other (5 ms of dependent launches squeezed between the encoder's grids); now they run beside each other as well
        (GeneralConditioner, NK_TE_OVERLAP) and the chain is short enough to hide.  Not when the towers or the encoder are replayed from
        hipGraphs (a replay belongs to the stream it was captured on)."""
        inputs = self.get_input(batch)
        cond = None
        from ..graphs import graphs_enabled

        if os.environ.get("NK_COND_OVERLAP", "1") != "0" and inputs.is_cuda and not (graphs_enabled("te") or graphs_enabled("vae")):
            main = torch.cuda.current_stream()
            if getattr(self, "_cond_stream", None) is None:
                self._cond_stream = torch.cuda.Stream(device=inputs.device)
            side = self._cond_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cond = self.conditioner(batch)
            latents = self.encode_first_stage(inputs)
            main.wait_stream(side)
            for t in cond.values():
                if torch.is_tensor(t):
                    t.record_stream(main)
        else:
            latents = self.encode_first_stage(inputs)
        batch["global_step"] = self.global_step
        loss = self(latents, batch, return_dict=False, cond=cond, **inject)
        self.last_log = {"train/loss": loss.detach().mean(), "train/loss_s0": loss.detach()[0]}
        return loss.mean()

    def accumulate(self, micro_batch_index: int, dp=None, last: bool = True):
        """Gradient accumulation (Lightning's `accumulate_grad_batches`, configs/sdxl/sdxl.example.yaml): call before the
        FORWARD of micro-batch `micro_batch_index` of an optimizer step (the overwrite / add mode is part of the replayed chain's signature:
        changing it between a forward and its backward raises).  The first micro-batch overwrites the gradients,
        later ones add; with a FlatDataParallel `dp`, only the last micro-batch exchanges them (DDP's no_sync)."""
        self.store.state.grad_accumulate = micro_batch_index > 0
        self._last_micro_batch = bool(last)
        if dp is not None:
            dp.no_sync(not last)

    def configure_adafactor(self, **kwargs):
        """Use the fused multi-tensor Adafactor (reference optimizers/adafactor.py; the optimizer the example configs name,
        configs/sdxl/sdxl.example.yaml:158-169) for optimizer_step().  kwargs as the reference class's; returns it."""
        if self.store is None:
            raise RuntimeError("call setup_flat_params() first")
        from ..optim import FlatAdafactor

        # block-aligned chunks only when the update is streamed behind backward: otherwise few, large chunks are faster
        kwargs.setdefault("boundaries", self._block_boundaries() if self.stream_optimizer else None)
        self.adafactor = FlatAdafactor(self.store, **kwargs)
        return self.adafactor

    def configure_ema(self, decay: float = 0.9999, use_num_updates: bool = True):
        """`use_ema` / `ema_decay_rate` of the reference engine (models/diffusion.py:47-48,93-99): a flat fp32 average of the
        trainable parameters, updated after every optimizer step (reference: on_train_batch_end, :243-244)."""
        if self.store is None:
            raise RuntimeError("call setup_flat_params() first")
        from ..optim import FlatEma

        # names as LitEma sees them: relative to the OpenAIWrapper (models/diffusion.py:96, ema.py:23-29)
        by_id = {id(p): n for n, p in self.model.named_parameters()}
        self.model_ema = FlatEma(self.store, decay, use_num_updates, names=[by_id[id(p)] for p in self.store.params])
        return self.model_ema

    def optimizer_step(self, lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, grad_scale: float = 1.0,
                       dp=None) -> None:
        """One parameter update on the flat buffers (gradients are not cleared: the next backward overwrites them): the configured Adafactor if
        configure_adafactor() was called (its own hyper-parameters; only grad_scale is used), else fused flat AdamW.

        The update is HBM-bound (2.6 G parameters: ~13 ms) and the next step begins with ~30 ms that never touch the UNet's weights
        (frozen VAE encoder: MFMA-bound convolutions; frozen conditioner: launch latency).  So the update is issued on a second stream
        behind the backward, and the main stream only waits for it where the UNet forward starts (`join_optimizer`, called by
        `forward`, `sample`, `ema_scope` and `state_dict`): the two overlap instead of queueing.  NK_OPT_OVERLAP=0 keeps it in line.

        dp: the FlatDataParallel wrapper when its exchange is SHARDED (mode rs_ag): this rank's optimizer then updates its own shard only
        (dp.attach_optimizer) and the ranks' new bf16 shadows are gathered right behind the update, on the same stream."""
        if self.store is None:
            raise RuntimeError("call setup_flat_params() first")
        sharded = dp is not None and getattr(dp, "sharded", False)
        if sharded and (getattr(self, "model_ema", None) is not None or getattr(self, "adafactor", None) is None or self._streaming_step):
            raise NotImplementedError("the sharded exchange (NK_DP_MODE=rs_ag) needs the fused Adafactor and supports neither EMA (its update reads "
                                      "every fp32 master) nor the streamed update")
        overlap = self.overlap_optimizer and self.store.master.is_cuda
        scope = nullcontext()
        if overlap:
            self.join_optimizer()
            if self._optimizer_stream is None:
                self._optimizer_stream = torch.cuda.Stream(device=self.store.master.device)
            self._optimizer_stream.wait_stream(torch.cuda.current_stream())      # gradients (and their exchange) are complete
            scope = torch.cuda.stream(self._optimizer_stream)
        with scope:
            opt = self._torch_optimizer
            if self._streaming_step:                                  # most of the update already went out behind backward
                if grad_scale != 1.0:
                    raise RuntimeError("a streamed optimizer step cannot take a grad_scale: chunks were already applied with 1.0")
                self.adafactor.end_step()
                self._streaming_step = False
                if opt is not None:
                    opt._step_count = getattr(opt, "_step_count", 0) + 1       # what torch's LRScheduler looks at
            elif opt is not None:
                opt.step(grad_scale=grad_scale)                      # the config's optimizer object (fused underneath)
            elif getattr(self, "adafactor", None) is not None:
                self.adafactor.step(grad_scale)
            else:
                self.store.adamw_step(lr, betas, eps, weight_decay, grad_scale)
            if self._torch_scheduler is not None:
                self._torch_scheduler.step()
            if getattr(self, "model_ema", None) is not None:
                self.model_ema.update()
            if sharded:
                dp.after_optimizer_step()      # every rank's shard of the new shadows, gathered behind the update
        self._optimizer_in_flight = overlap
        self.store.state.grad_accumulate = False
        self.global_step += 1

    def join_optimizer(self) -> None:
        """Make the current stream wait for a parameter update still running on the optimizer stream."""
        if self._optimizer_in_flight:
            torch.cuda.current_stream().wait_stream(self._optimizer_stream)
            self._optimizer_in_flight = False

    def state_dict(self, *args, **kwargs):
        """Parameters as the reference's keys name them.  Under the sharded data-parallel exchange (dp.FlatDataParallel, mode rs_ag) the
        fp32 masters of the other ranks' parts are stale between checkpoints.  Gathering them is a COLLECTIVE (dp.sync_masters()), and a
        state_dict() call is often rank-local (`if rank == 0: torch.save(engine.state_dict())`, EMA / log_images tooling): an implicit
        collective here would deadlock those.  So this never communicates: it raises unless the masters are whole --
        call `engine.sync_masters()` on EVERY rank first (trainer.lightning.DiffusionEngineMI355X.state_dict(), which Lightning calls on every
        rank at checkpoint time, does exactly that and says so)."""
        self.join_optimizer()
        dp = getattr(getattr(self, "store", None), "dp", None)
        if dp is not None and dp.sharded and not dp.masters_whole:
            raise RuntimeError("DiffusionEngine.state_dict(): the fp32 masters are sharded over the ranks (NK_DP_MODE=rs_ag) and stale for foreign "
                               "parts; call engine.sync_masters() on every rank before any rank reads the state dict")
        return super().state_dict(*args, **kwargs)

    def sync_masters(self) -> None:
        """rs_ag: make every rank's fp32 masters and optimizer statistics whole (a collective: all ranks).  No-op otherwise."""
        self.join_optimizer()
        dp = getattr(getattr(self, "store", None), "dp", None)
        if dp is not None and dp.sharded:
            dp.sync_masters()

    # -- sampling -------------------------------------------------------------------
    @contextmanager
    def ema_scope(self, context: Optional[str] = None):
        """models/diffusion.py:280-292: run the body with the EMA weights swapped in (no-op without EMA)."""
        ema = getattr(self, "model_ema", None) if self.use_ema else None
        self.join_optimizer()
        if ema is not None:
            ema.store()
            ema.copy_to()
        try:
            yield None
        finally:
            if ema is not None:
                ema.restore()

    @torch.no_grad()
    def sample(self, cond: dict, uc: Optional[dict] = None, batch_size: int = 4, shape=None, noise: Optional[Tensor] = None, **model_kwargs) -> Tensor:
        """models/diffusion.py:298-313.  `noise` may be injected (parity tests); otherwise unit gaussian of `shape`."""
        if self.sampler is None:
            raise RuntimeError("no sampler configured")
        from ..modules.diffusion.sampling import FusedDenoiser

        self.join_optimizer()

        device = next(self.model.parameters()).device
        randn = torch.randn(batch_size, *shape, device=device) if noise is None else noise.to(device=device, dtype=torch.float32).clone()
        if model_kwargs:
            fused = FusedDenoiser(self.model, self.denoiser, **model_kwargs)
        else:   # kept between calls: it owns the captured hipGraph of a sampling step
            fused = self.__dict__.setdefault("_fused_denoiser", FusedDenoiser(self.model, self.denoiser))
        return self.sampler(fused, randn, cond, uc=uc).clone()

    @torch.no_grad()
    def log_images(self, batch: dict, num_img: int = 4, split: str = "train", sample: bool = True, ucg_keys=None, **kwargs) -> dict:
        """models/diffusion.py:369-420: inputs, VAE reconstructions and (CFG) samples for the batch's conditioning."""
        inputs = self.get_input(batch)[:num_img]
        num_img = len(inputs)
        input_keys = list({e.input_key for e in getattr(self.conditioner, "embedders", ()) if hasattr(e, "input_key")})
        if ucg_keys and any(k not in input_keys for k in ucg_keys):
            raise ValueError("Each defined ucg key for sampling must be in the provided conditioner input keys!"
                             f"\nRequested UCG keys: {ucg_keys}\nAvailable input keys: {input_keys}")
        latents = self.encode_first_stage(inputs)
        images = {f"{split}/inputs": inputs.cpu(), f"{split}/recons": self.decode_first_stage(latents).cpu()}
        cond, uncond = get_unconditional_conditioning(self.conditioner, batch)
        device = latents.device
        for key, value in cond.items():
            if isinstance(value, Tensor):
                cond[key], uncond[key] = value[:num_img].to(device), uncond[key][:num_img].to(device)
        if sample:
            with self.ema_scope("Plotting"):
                samples = self.sample(cond=cond, shape=latents.shape[1:], uc=uncond, batch_size=num_img, **kwargs)
            images["samples"] = self.decode_first_stage(samples).cpu()
        return images


def get_unconditional_conditioning(conditioner, batch_c: dict, batch_uc: Optional[dict] = None, force_uc_zero_embeddings=None,
                                   force_cond_zero_embeddings=None):
    """models/diffusion.py:423-447: (cond, uncond) with every embedder's ucg dropout switched off for the two calls; the
    unconditional batch defaults to the same batch with empty captions."""
    embedders = list(getattr(conditioner, "embedders", ()))
    rates = [e.ucg_rate for e in embedders]
    for e in embedders:
        e.ucg_rate = 0.0
    try:
        c = conditioner(batch_c, force_zero_embeddings=force_cond_zero_embeddings)
        if batch_uc is None:
            batch_uc = dict(batch_c)
            batch_uc["caption"] = [""] * len(batch_c["caption"]) if "caption" in batch_c else [""]
        uc = conditioner(batch_uc, force_zero_embeddings=force_uc_zero_embeddings or [])
    finally:
        for e, rate in zip(embedders, rates):
            e.ucg_rate = rate
    return c, uc
