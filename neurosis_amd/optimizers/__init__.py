"""MI355X mirror of `neurosis.optimizers` (/root/reference/src/neurosis/optimizers/__init__.py): the optimizer block of the
example configs (`configs/sdxl/sdxl.example.yaml:158-169`, `configs/sd15/sd15.example.yml`) names
`neurosis.optimizers.Adafactor` and `neurosis.optimizers.AdafactorScheduler`; under the `neurosis.` -> `neurosis_amd.` prefix
swap these resolve here.

`Adafactor` keeps the reference class's constructor (`optimizers/adafactor.py:100-131`) and is a real
`torch.optim.Optimizer` (so LightningCLI's `OptimizerCallable`, `configure_optimizers` and checkpointing accept it), but its
`step()` is the fused multi-tensor update of `neurosis_amd.optim.FlatAdafactor` on the flat fp32 master / gradient
buffers: a handful of HIP launches for the whole UNet instead of a Python loop over ~1 700 tensors.  There is no eager
fallback: parameters must live in a `FlatParamStore` (the engine's `setup_flat_params()` puts them there; parameters
handed over on a GPU without one are re-homed into a new store on the first step).

`AdamW` is the fused flat AdamW (`nk_adamw_flat`) under `torch.optim.AdamW`'s constructor: not named by the reference's
configs, provided because "any subclass of torch.optim.Optimizer" is what its YAML comment invites.
`HybridOptimizer` / `HybridScheduler` (one optimizer per parameter group) are outside the SD/SDXL example configs and are
not built.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

from ..optim import FlatAdafactor

__all__ = ["Adafactor", "AdafactorScheduler", "AdamW"]


def _group_store(group: dict, who: str):
    """The FlatParamStore that holds exactly this group's parameters (created on a GPU if they have none yet)."""
    from ..nn import FlatParamStore

    params = [p for p in group["params"] if p.requires_grad]
    if not params:
        raise ValueError(f"{who}: a parameter group without trainable parameters")
    stores = {id(getattr(p, "_nk_store", None)): getattr(p, "_nk_store", None) for p in params}
    if None in stores.values():
        if len(stores) > 1:
            raise ValueError(f"{who}: a parameter group mixes store-managed and free parameters")
        if not params[0].is_cuda:
            raise RuntimeError(f"{who}: the fused update runs on HIP buffers; move the model to the GPU (and call "
                               "setup_flat_params()) before the first step -- there is no CPU path")
        return FlatParamStore(params)
    if len(stores) != 1:
        raise ValueError(f"{who}: the parameters of one group live in {len(stores)} different flat stores")
    store = next(iter(stores.values()))
    if len(store.params) != len(params) or any(a is not b for a, b in zip(store.params, params)):
        raise ValueError(f"{who}: a parameter group must cover its flat store exactly ({len(params)} parameters given, "
                         f"{len(store.params)} in the store): the fused kernels update the whole buffer")
    return store


class Adafactor(Optimizer):
    """`neurosis.optimizers.Adafactor` (reference optimizers/adafactor.py:100-255), fused."""

    def __init__(self, params, lr: Optional[float] = None, eps: tuple[float, float] = (1e-30, 1e-3), clip_threshold: float = 1.0,
                 decay_rate: float = -0.8, beta1: Optional[float] = None, weight_decay: float = 0.0, scale_parameter: bool = True,
                 relative_step: bool = True, warmup_init: bool = False):
        if lr is not None and relative_step:
            raise ValueError("Cannot combine manual `lr` and `relative_step=True` options")
        if warmup_init and not relative_step:
            raise ValueError("`warmup_init=True` requires `relative_step=True`")
        if beta1 is not None:
            raise NotImplementedError("neurosis_amd.optimizers.Adafactor: beta1 (first moment) is not fused; the reference configs use beta1=None")
        defaults = dict(lr=lr, eps=eps, clip_threshold=clip_threshold, decay_rate=decay_rate, beta1=beta1, weight_decay=weight_decay,
                        scale_parameter=scale_parameter, relative_step=relative_step, warmup_init=warmup_init, differentiable=False)
        super().__init__(params, defaults)
        self._flat: list[FlatAdafactor] = []
        self._pending_state: Optional[dict] = None

    # -- binding to the flat buffers --------------------------------------------------------------------
    def bind(self) -> list[FlatAdafactor]:
        """One FlatAdafactor per parameter group (each group = one flat store).  Idempotent."""
        if not self._flat:
            for g in self.param_groups:
                store = _group_store(g, "Adafactor")
                self._flat.append(FlatAdafactor(store, lr=g["lr"] if not g["relative_step"] else None, eps=tuple(g["eps"]),
                                                clip_threshold=g["clip_threshold"], decay_rate=g["decay_rate"], beta1=g["beta1"],
                                                weight_decay=g["weight_decay"], scale_parameter=g["scale_parameter"],
                                                relative_step=g["relative_step"], warmup_init=g["warmup_init"],
                                                boundaries=g.get("chunk_boundaries")))
            if self._pending_state is not None:
                sd, self._pending_state = self._pending_state, None
                self._load_flat(sd)
        return self._flat

    @property
    def flat(self) -> FlatAdafactor:
        """The fused optimizer of the first (UNet) group."""
        return self.bind()[0]

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for f in self.bind():
            f.step(grad_scale)
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        """Nothing to do, on purpose: `.grad` tensors are views of the store's flat gradient buffer and every gradient is
        OVERWRITTEN by the kernel that produces it on the first micro-batch of a step (FlatParamStore docstring); setting
        them to None -- torch's default -- would detach the parameters from that buffer."""

    @staticmethod
    def _get_lr(param_group: dict, param_state: dict) -> float:
        """adafactor.py:133-147 (what AdafactorScheduler reports), from the group's hyper-parameters and a state holding
        `step` and `RMS`."""
        import math

        rel_step_sz = param_group["lr"]
        if param_group["relative_step"]:
            min_step = 1e-6 * param_state["step"] if param_group["warmup_init"] else 1e-2
            rel_step_sz = min(min_step, 1.0 / math.sqrt(param_state["step"]))
        param_scale = 1.0
        if param_group["scale_parameter"]:
            param_scale = max(param_group["eps"][1], float(param_state["RMS"]))
        return param_scale * rel_step_sz

    # -- checkpointing: torch's layout, the reference's per-parameter keys ------------------------------
    def state_dict(self) -> dict:
        groups, state, base = [], {}, 0
        flats = self._flat
        for gi, g in enumerate(self.param_groups):
            n = len(g["params"])
            packed = {k: v for k, v in g.items() if k != "params"}
            packed["params"] = list(range(base, base + n))
            groups.append(packed)
            if gi < len(flats):
                for i, st in flats[gi].state_dict()["state"].items():
                    state[base + i] = st
            base += n
        if not flats and self._pending_state is not None:
            state = self._pending_state["state"]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict: dict) -> None:
        for g, saved in zip(self.param_groups, state_dict.get("param_groups", [])):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
        if self._flat:
            self._load_flat(state_dict)
        else:
            self._pending_state = state_dict      # applied when the flat buffers exist (first step / bind())

    def _load_flat(self, sd: dict) -> None:
        base = 0
        for g, f in zip(self.param_groups, self._flat):
            n = len(g["params"])
            f.load_state_dict({"state": {int(i) - base: st for i, st in sd.get("state", {}).items() if base <= int(i) < base + n}})
            base += n


class AdafactorScheduler(LambdaLR):
    """`neurosis.optimizers.AdafactorScheduler` (adafactor.py:258-291): a proxy that reports `initial_lr` before the first
    step and the optimizer's own per-group lr afterwards (here read from the fused kernel's per-tensor lr table)."""

    def __init__(self, optimizer: Optimizer, initial_lr: float = 0.0):
        self.initial_lr = initial_lr

        def lr_lambda(_):
            return self.initial_lr

        for group in optimizer.param_groups:
            group["initial_lr"] = initial_lr
        super().__init__(optimizer, lr_lambda)
        for group in optimizer.param_groups:
            del group["initial_lr"]

    def get_lr(self):
        opt = self.optimizer
        flats = getattr(opt, "_flat", [])
        lrs = [float(f.lr_t[0]) for f in flats if f.step_count > 0]
        if len(lrs) == 0:
            lrs = self.base_lrs  # if called before stepping
        return lrs


class AdamW(Optimizer):
    """Fused flat AdamW (`FlatParamStore.adamw_step`, one launch over all parameters) under `torch.optim.AdamW`'s arguments."""

    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._stores = []

    def bind(self):
        if not self._stores:
            self._stores = [_group_store(g, "AdamW") for g in self.param_groups]
        return self._stores

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for g, store in zip(self.param_groups, self.bind()):
            store.adamw_step(g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"], grad_scale)
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        """See Adafactor.zero_grad: gradients are overwritten by their producers; the views must stay attached."""

    def state_dict(self) -> dict:
        groups, state, base = [], {}, 0
        for gi, g in enumerate(self.param_groups):
            n = len(g["params"])
            groups.append({**{k: v for k, v in g.items() if k != "params"}, "params": list(range(base, base + n))})
            if gi < len(self._stores):
                for i, st in self._stores[gi].optimizer_state_dict()["state"].items():
                    state[base + i] = st
            base += n
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict: dict) -> None:
        for g, saved in zip(self.param_groups, state_dict.get("param_groups", [])):
            for k, v in saved.items():
                if k != "params":
                    g[k] = v
        base = 0
        for g, store in zip(self.param_groups, self.bind()):
            n = len(g["params"])
            store.load_optimizer_state_dict({"state": {int(i) - base: st for i, st in state_dict.get("state", {}).items() if base <= int(i) < base + n}})
            base += n
