"""Lightning adapter: keeps the reference's Trainer / LightningCLI in charge of the loop while the step itself runs on the
MI355X engine.

The reference's `DiffusionEngine` IS a `LightningModule` (/root/reference/src/neurosis/models/diffusion.py:35) trained under
Lightning's `bf16-mixed` autocast, automatic optimization, `DDPStrategy` and `accumulate_grad_batches: 4`
(configs/sdxl/sdxl.example.yaml:3-15).  Here those four become: the engine's own precision policy, manual
optimization (the gradients live in one flat fp32 buffer and the update is a few fused launches), `FlatDataParallel` (flat
RCCL all-reduce overlapped with the backward tail) and `DiffusionEngine.accumulate` (first micro-batch overwrites, later ones
add, only the last one exchanges).  Constructor arguments = the reference engine's, so the YAML `model:` block is unchanged
apart from the class path:

    model:
      class_path: neurosis_amd.trainer.DiffusionEngineMI355X
      init_args: { ...the init_args of neurosis.models.DiffusionEngine, class paths prefix-swapped... }

`lightning` is not installed in the build image: the import is guarded, the module always imports, and instantiating the
adapter without Lightning raises an ImportError that says so (tests exercise the step logic through a stand-in trainer).
"""
from __future__ import annotations

from typing import Any, Optional

import torch

from ..models.diffusion import DiffusionEngine

try:  # pragma: no cover - depends on the environment
    import lightning.pytorch as _L

    _Base = _L.LightningModule
    _HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001 - any failure to import means "not available"
    _Base = torch.nn.Module
    _HAVE_LIGHTNING = False


def lightning_available() -> bool:
    return _HAVE_LIGHTNING


class DiffusionEngineMI355X(_Base):
    """`neurosis.models.DiffusionEngine` for Lightning, on the HIP engine.  `require_lightning=False` lets tests drive
    `on_fit_start / training_step` with a stand-in trainer object."""

    def __init__(self, require_lightning: bool = True, accumulate_grad_batches: Optional[int] = None, wire_dtype: Optional[str] = None, **engine_kwargs):
        if require_lightning and not _HAVE_LIGHTNING:
            raise ImportError("neurosis_amd.trainer.DiffusionEngineMI355X needs `lightning` (>= 2.2.1, the reference's pin); it is not importable here")
        super().__init__()
        self.automatic_optimization = False      # gradients: one flat fp32 buffer; update: fused kernels; exchange: FlatDataParallel
        self.engine = DiffusionEngine(**engine_kwargs)
        self._accumulate = accumulate_grad_batches
        self._wire_dtype = {None: None, "fp32": None, "bf16": torch.bfloat16}[wire_dtype]
        self.dp = None
        self._micro = 0
        # Lightning restores a checkpoint (load_state_dict, on_load_checkpoint) BEFORE on_fit_start, i.e. before the flat store, the
        # fused optimizer and the EMA exist: what arrives early is parked here and applied at the end of on_fit_start
        self._pending_optimizer: Optional[dict] = None
        self._pending_ema: Optional[dict] = None

    # -- setup --------------------------------------------------------------------------------------------
    def _trainer_attr(self, name: str, default: Any) -> Any:
        tr = getattr(self, "_trainer_stub", None) or (getattr(self, "trainer", None) if _HAVE_LIGHTNING else None)
        return getattr(tr, name, default) if tr is not None else default

    def on_fit_start(self) -> None:
        from ..dp import FlatDataParallel

        dev = self._trainer_attr("device", None) or getattr(self, "device", None) or next(self.engine.parameters()).device
        if isinstance(dev, torch.device) and dev.type == "cuda":
            self.engine.to(dev)
        self.engine.setup_flat_params()          # builds the config's optimizer / scheduler too (configure_optimizers)
        if self._accumulate is None:
            self._accumulate = int(self._trainer_attr("accumulate_grad_batches", 1) or 1)
        world = int(self._trainer_attr("world_size", 1) or 1)
        if world > 1:
            self.dp = FlatDataParallel(self.engine.model.diffusion_model, self.engine.store, wire_dtype=self._wire_dtype)
        self._apply_pending()
        if self.dp is not None and self.dp.sharded:        # NK_DP_MODE=rs_ag: this rank updates its shard only
            if getattr(self.engine, "adafactor", None) is None:
                raise NotImplementedError("NK_DP_MODE=rs_ag needs the fused Adafactor (the optimizer that can be restricted to a shard)")
            self.dp.attach_optimizer(self.engine.adafactor)

    def _apply_pending(self) -> None:
        """checkpoint state that arrived before the store / optimizer / EMA existed (Trainer.fit(ckpt_path=...))"""
        if self._pending_ema is not None:
            ema = getattr(self.engine, "model_ema", None)
            if ema is None:
                raise RuntimeError("the checkpoint carries engine.model_ema.* but this engine was built with use_ema=False")
            ema.load_state_dict(self._pending_ema)
            self._pending_ema = None
        if self._pending_optimizer is not None:
            opt = self.engine._torch_optimizer
            if opt is None:
                raise RuntimeError("the checkpoint carries optimizer state (nk_optimizer) but the engine has no optimizer configured")
            opt.load_state_dict(self._pending_optimizer)
            self._pending_optimizer = None
            self.engine.store.masters_changed()

    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        """`engine.model_ema.*` entries are held back while the EMA module does not exist yet (it is created with the flat store in
        on_fit_start): a strict load would otherwise report them as unexpected keys."""
        prefix = "engine.model_ema."
        if getattr(self.engine, "model_ema", None) is None and self.engine.use_ema:
            held = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
            if held:
                self._pending_ema = held
                state_dict = {k: v for k, v in state_dict.items() if not k.startswith(prefix)}
        return super().load_state_dict(state_dict, strict=strict, **kwargs)

    # -- the step -----------------------------------------------------------------------------------------
    def training_step(self, batch: dict, batch_idx: int):
        acc = max(int(self._accumulate or 1), 1)
        last = self._micro == acc - 1
        self.engine.accumulate(self._micro, self.dp, last=last)
        loss = self.engine.training_step(batch, batch_idx)
        (loss / acc).backward()
        if last:
            scale = self.dp.finish() if self.dp is not None else 1.0
            self.engine.optimizer_step(grad_scale=scale, dp=self.dp)
            self._micro = 0
        else:
            self._micro += 1
        if _HAVE_LIGHTNING and getattr(self, "_trainer_stub", None) is None:
            self.log_dict(self.engine.last_log, prog_bar=True, on_step=True, on_epoch=False)
        return loss.detach()

    def configure_optimizers(self):
        """Manual optimization: Lightning gets no optimizer to step; the engine's fused optimizer checkpoints through
        `on_save_checkpoint` / `on_load_checkpoint` below."""
        return None

    # -- checkpoints: the reference's state_dict keys live under `engine.`; optimizer state rides along ------------------
    collective_state_dict = True      # False: state_dict() never communicates; under rs_ag it raises while the masters are sharded (as the engine's)

    def state_dict(self, *args, **kwargs):
        """COLLECTIVE under the sharded exchange (dp mode rs_ag): every rank must call it.  Lightning's dump_checkpoint does -- it builds the
        module's state_dict on EVERY rank and only rank 0 writes the file -- and Lightning offers no earlier all-rank hook (on_save_checkpoint
        runs after the state_dict is built), so the sharded fp32 masters / optimizer statistics are made whole HERE (engine.sync_masters();
        engine.state_dict() itself never communicates, and raises while they are not whole).  A rank-LOCAL call -- a rank-zero callback, an export
        script, an EMA dump -- would therefore hang in the collective: such code either calls `engine.sync_masters()` on every rank first, or sets
        `collective_state_dict = False`, which turns the hidden collective into the engine's explicit error.  All-reduce mode (the default)
        never communicates here."""
        if self.collective_state_dict:
            self.engine.sync_masters()
        return super().state_dict(*args, **kwargs)

    def on_save_checkpoint(self, checkpoint: dict) -> None:
        self.engine.join_optimizer()
        # (rs_ag: state_dict() above has already made every rank's masters and optimizer statistics whole -- Lightning builds the
        # module's state_dict before this hook runs -- so what is saved below is complete whether or not the tensors alias the flat buffers)
        opt = self.engine._torch_optimizer
        if opt is not None:
            checkpoint["nk_optimizer"] = opt.state_dict()
        checkpoint["nk_global_step"] = self.engine.global_step

    def on_load_checkpoint(self, checkpoint: dict) -> None:
        opt = self.engine._torch_optimizer
        if "nk_optimizer" in checkpoint:
            if opt is not None:
                opt.load_state_dict(checkpoint["nk_optimizer"])
            else:                       # the usual order under Trainer.fit(ckpt_path=...): applied by on_fit_start
                self._pending_optimizer = checkpoint["nk_optimizer"]
        self.engine.global_step = int(checkpoint.get("nk_global_step", self.engine.global_step))
        if self.engine.store is not None:
            self.engine.store.masters_changed()

    # -- pass-throughs the reference's callbacks use (image logger, EMA scope) ---------------------------------------------
    def log_images(self, batch: dict, **kwargs) -> dict:
        return self.engine.log_images(batch, **kwargs)

    def ema_scope(self, context: Optional[str] = None):
        return self.engine.ema_scope(context)
