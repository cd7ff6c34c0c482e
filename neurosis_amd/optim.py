"""Optimizers over the flat parameter store.

`FlatAdafactor` mirrors `neurosis.optimizers.Adafactor` (reference `optimizers/adafactor.py:104-255`: same constructor
arguments and the same per-tensor rule -- factored second moments over the LAST TWO dims of every tensor with >= 2 dims,
RMS-scaled relative step, update clipping) but runs as a handful of multi-tensor HIP launches on the flat fp32 master /
gradient buffers (`csrc/optim.hip`), rewriting the bf16 shadows in the same pass.  `AdafactorScheduler` mirrors :258-291.

Differences, all deliberate: `beta1` (first moment) is not implemented -- the example configs leave it `None`
(`configs/sdxl/sdxl.example.yaml:158-164`) -- and asking for it raises; state lives in two flat buffers instead of a
per-parameter dict; `state_dict()` / `load_state_dict()` speak the reference optimizer's checkpoint format (per-parameter
`step`, `exp_avg_sq_row`, `exp_avg_sq_col` / `exp_avg_sq`, `RMS`) so a run resumes with its second moments, step count
(relative-step warm-up, beta2_t) and per-tensor RMS intact.  `FlatEma` is an `nn.Module` carrying `LitEma`'s buffer names
(`decay`, `num_updates`, one shadow per parameter), so `model_ema.*` keys of a reference checkpoint load and save unchanged.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch
from torch import Tensor

from . import ops
from .lib import call, query

AF_TENSOR_DTYPE = np.dtype([("off", "<i8"), ("row_off", "<i8"), ("col_off", "<i8"), ("ws_row", "<i8"), ("ws_col", "<i8"),
                            ("kind", "<i4"), ("d0", "<i4"), ("d1", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("item0", "<i4"),
                            ("nitems", "<i4"), ("mr0", "<i4"), ("cnt0", "<i4"), ("pad", "<i4")])
AF_ITEM_DTYPE = np.dtype([("tensor", "<i4"), ("tr", "<i4"), ("tc", "<i4"), ("pad", "<i4")])
AF_TR, AF_TC, AF_CONV_PAIRS, AF_VEC = 256, 64, 1024, 1024


class _Args(C.Structure):
    _fields_ = [("master", C.c_void_p), ("grad", C.c_void_p), ("shadow", C.c_void_p), ("state", C.c_void_p), ("ws", C.c_void_p),
                ("tensors", C.c_void_p), ("items", C.c_void_p), ("u2_part", C.c_void_p), ("p2_part", C.c_void_p),
                ("mean_row", C.c_void_p), ("scale", C.c_void_p), ("lr_t", C.c_void_p),
                ("item_lo", C.c_int), ("item_hi", C.c_int), ("tensor_lo", C.c_int), ("tensor_hi", C.c_int),
                ("beta2t", C.c_float), ("eps1", C.c_float), ("eps2", C.c_float), ("clip_threshold", C.c_float),
                ("rel_step", C.c_float), ("weight_decay", C.c_float), ("grad_scale", C.c_float),
                ("scale_parameter", C.c_int), ("counters", C.c_void_p), ("has_matrix", C.c_int), ("reserved", C.c_int)]


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class FlatAdafactor:
    """Adafactor on a FlatParamStore.  Constructor arguments as the reference's (adafactor.py:104-131)."""

    def __init__(self, store, lr: Optional[float] = None, eps: tuple[float, float] = (1e-30, 1e-3), clip_threshold: float = 1.0,
                 decay_rate: float = -0.8, beta1: Optional[float] = None, weight_decay: float = 0.0, scale_parameter: bool = True,
                 relative_step: bool = True, warmup_init: bool = False, chunk_bytes: Optional[int] = None, boundaries=None):
        if lr is not None and relative_step:
            raise ValueError("Cannot combine manual `lr` and `relative_step=True` options")
        if warmup_init and not relative_step:
            raise ValueError("`warmup_init=True` requires `relative_step=True`")
        if beta1 is not None:
            raise NotImplementedError("FlatAdafactor: beta1 (first moment) is not implemented; the reference configs use beta1=None")
        if query("nk_adafactor_tensor_bytes") != AF_TENSOR_DTYPE.itemsize:
            raise RuntimeError("FlatAdafactor: tensor table layout differs from the HIP library's")
        if chunk_bytes is None:
            # gradient bytes per chunk of tensors (three launches each).  A chunk's gradients are re-read (second-moment pass, update-RMS
            # pass, apply pass) out of the 256 MB Infinity Cache instead of HBM while the chunk fits it -- 14 instead of 22 HBM bytes per
            # parameter.  The update runs on its own stream beside the next step's VAE encoder, so what counts is the HBM bandwidth it takes
            # from that encoder: round 3 (five launches per chunk) measured a steady-state step of 172.2 / 172.8 / 173.8 / 174.8 ms at
            # 64 / 128 / 256 / 2048 MB; round 4 (three launches per chunk): step-time p50 164.3 / 165.2 / 165.5 ms at 64 / 128 / 256 MB
            # (one box, back to back) -- the cache-sized chunk still wins although it costs twice the launches.
            import os

            # 128 MB (round 5; 64 before): alone on the chip the update takes 20.5 / 17.8 / 13.8 / 12.7 / 12.2 / 11.7 ms at 32 / 64 / 128 / 256 /
            # 512 / 1024 MB chunks (tools/bench_optimizer.py) -- three dependent launches per chunk, each with its ramp and tail -- while the
            # second and third read of a chunk's gradients only stay in the 256 MB Infinity Cache for small chunks; in the step 128 MB measured
            # -0.5 ms against 64, 256 the same as 64, 512 +0.6 ms (profiles/r05_ab_notes.txt)
            chunk_bytes = int(os.environ.get("NK_AF_CHUNK_MB", "128")) << 20
        # `boundaries`: tensor indices at which a new chunk must begin (the first parameter of every top-level UNet block), so
        # that a block's update can be issued as soon as that block's gradients are final, while backward is still running
        bounds = set(int(b) for b in (boundaries or ()))
        self.store = store
        self.lr, self.eps, self.clip_threshold, self.decay_rate = lr, eps, clip_threshold, decay_rate
        self.weight_decay, self.scale_parameter, self.relative_step, self.warmup_init = weight_decay, scale_parameter, relative_step, warmup_init
        self.step_count = 0
        dev = store.master.device

        tens = np.zeros(len(store.params), dtype=AF_TENSOR_DTYPE)
        items = []
        self.chunks = []   # (tensor_lo, tensor_hi, item_lo, item_hi)
        state_off = 0
        ws_max = 0
        mr_slots = 0
        n_counters = 0
        c_t0, c_i0, c_bytes, c_ws = 0, 0, 0, 0
        for ti, (p, off) in enumerate(zip(store.params, store.offsets)):
            t = tens[ti]
            t["off"] = off
            t["cnt0"] = n_counters
            if p.dim() >= 2 and p.dim() not in (2, 4):
                raise NotImplementedError(f"FlatAdafactor: {p.dim()}-d parameters are not supported")
            nbytes = p.numel() * 4
            if c_bytes and (c_bytes + nbytes > chunk_bytes or ti in bounds):   # close the current chunk before this tensor
                self.chunks.append((c_t0, ti, c_i0, len(items)))
                ws_max = max(ws_max, c_ws)
                c_t0, c_i0, c_bytes, c_ws = ti, len(items), 0, 0
            t["item0"] = len(items)
            if p.dim() == 2:
                d0, d1 = p.shape
                if d1 % 4:
                    raise NotImplementedError("FlatAdafactor: matrix rows must be a multiple of 4 elements")
                ntr, ntc = -(-d0 // AF_TR), -(-d1 // AF_TC)
                t["kind"], t["d0"], t["d1"], t["kh"], t["kw"] = 1, d0, d1, 1, 1
                t["row_off"], t["col_off"] = state_off, state_off + _align(d0)
                state_off += _align(d0) + _align(d1)
                t["ws_row"], t["ws_col"] = c_ws, c_ws + ntc * d0
                c_ws += _align(ntc * d0 + ntr * d1)
                items += [(ti, r, c, 0) for r in range(ntr) for c in range(ntc)]
                t["mr0"] = mr_slots
                mr_slots += ntr
                n_counters += ntr + ntc
            elif p.dim() == 4:
                O, I, KH, KW = p.shape
                if KH > 3 or KW > 3:
                    raise NotImplementedError("FlatAdafactor: conv kernels larger than 3x3 are not supported")
                t["kind"], t["d0"], t["d1"], t["kh"], t["kw"] = 2, O, I, KH, KW
                t["row_off"], t["col_off"] = state_off, state_off + _align(O * KH * I)
                state_off += _align(O * KH * I) + _align(O * KW * I)
                items += [(ti, r, 0, 0) for r in range(-(-(O * I) // AF_CONV_PAIRS))]
            else:
                n = p.numel()
                t["kind"], t["d0"], t["d1"], t["kh"], t["kw"] = 0, n, 1, 1, 1
                t["row_off"] = state_off
                state_off += _align(n)
                items += [(ti, r, 0, 0) for r in range(-(-n // AF_VEC))]
            t["nitems"] = len(items) - int(t["item0"])
            n_counters += 1
            c_bytes += nbytes
        self.chunks.append((c_t0, len(store.params), c_i0, len(items)))
        ws_max = max(ws_max, c_ws)

        items_np = np.array(items, dtype=np.int32).view(AF_ITEM_DTYPE).reshape(-1)
        self._tens_np = tens
        self.tensors = torch.from_numpy(tens.view(np.uint8).copy()).to(dev)
        self.items = torch.from_numpy(items_np.view(np.uint8).copy()).to(dev)
        # "blocks done" counters of the last-block-done finalisations (optim.hip): zeroed before every step (begin_step) although every pass
        # leaves them at zero -- a step abandoned half way by the health gate must not poison the next one
        self.counters = torch.zeros(n_counters, dtype=torch.int32, device=dev)
        self.nitems, self.ntensors = len(items), len(store.params)
        self.state = torch.zeros(max(state_off, 1), dtype=torch.float32, device=dev)
        # NK_AF_STREAMS (default 1; 2 = A/B): consecutive chunks alternate between two streams forked from the caller's, so that one chunk's
        # tail (counter round trips, finishing blocks) runs beside the next chunk's loads.  Chunks share nothing but the
        # partial-sum workspace, which then exists once per stream.
        import os as _os

        self.nstreams = 2 if _os.environ.get("NK_AF_STREAMS", "1") == "2" else 1
        self._side = None
        self.ws = torch.empty(max(ws_max, 1) * self.nstreams, dtype=torch.float32, device=dev)
        self._ws_stride = max(ws_max, 1)
        self.u2_part = torch.zeros(self.nitems, dtype=torch.float32, device=dev)
        self.p2_part = torch.zeros(self.nitems, dtype=torch.float32, device=dev)
        self.mean_row = torch.zeros(max(mr_slots, 1), dtype=torch.float32, device=dev)
        self.scale = torch.zeros(self.ntensors, dtype=torch.float32, device=dev)
        self.lr_t = torch.zeros(self.ntensors, dtype=torch.float32, device=dev)
        self._p2_valid = False
        self.owned = None          # restrict_ranges(): [(tensor_lo, tensor_hi), ...] this process updates; None = everything
        store.add_listener(self)

    def restrict(self, tensor_lo: int, tensor_hi: int) -> None:
        self.restrict_ranges([(tensor_lo, tensor_hi)])

    def restrict_ranges(self, ranges) -> None:
        """Update only the tensors inside `ranges` = [(tensor_lo, tensor_hi), ...] from now on (data-parallel rs_ag mode: a rank owns one
        tensor-aligned part of every slice of the flat buffers; the others' updated shadows arrive by all-gather).  Chunks are cut at every
        boundary; per-tensor state never crosses a tensor, so cutting changes no arithmetic."""
        ranges = sorted((int(a), int(b)) for a, b in ranges if b > a)
        bounds = sorted({x for ab in ranges for x in ab})
        cut = []
        for (t0, t1, i0, i1) in self.chunks:
            marks = sorted({t0, t1} | {b for b in bounds if t0 < b < t1})
            for a, b in zip(marks[:-1], marks[1:]):
                ia = i0 if a == t0 else int(self._tens_np[a]["item0"])
                ib = i1 if b == t1 else int(self._tens_np[b]["item0"])
                cut.append((a, b, ia, ib))
        self.chunks = cut
        self.owned = ranges

    def state_span(self, tensor_lo: int, tensor_hi: int) -> tuple[int, int]:
        """[lo, hi) of `self.state` (fp32 second-moment statistics) belonging to tensors [tensor_lo, tensor_hi): state is laid out in
        tensor order, so a tensor-aligned shard's statistics are one contiguous span (rs_ag gathers them for checkpoints)"""
        lo = int(self._tens_np[tensor_lo]["row_off"]) if tensor_lo < self.ntensors else self.state.numel()
        hi = int(self._tens_np[tensor_hi]["row_off"]) if tensor_hi < self.ntensors else self.state.numel()
        return lo, hi

    def _mine(self, ci: int) -> bool:
        return self.owned is None or any(self.chunks[ci][0] >= a and self.chunks[ci][1] <= b for a, b in self.owned)

    def masters_changed(self) -> None:
        """FlatParamStore hook: the fp32 masters were rewritten from outside (checkpoint load, broadcast, EMA swap): the
        per-tile sums of p^2 that scale_parameter uses are stale."""
        self._p2_valid = False

    # -- the update -------------------------------------------------------------------------------
    def _args(self, chunk, beta2t: float, rel_step: float, grad_scale: float, ws_slot: int = 0) -> _Args:
        t0, t1, i0, i1 = chunk
        s = self.store
        has_matrix = int((self._tens_np["kind"][t0:t1] == 1).any())
        return _Args(s.master.data_ptr(), s.grad.data_ptr(), s.shadow.data_ptr(), self.state.data_ptr(), self.ws.data_ptr() + 4 * self._ws_stride * ws_slot,
                     self.tensors.data_ptr(), self.items.data_ptr(), self.u2_part.data_ptr(), self.p2_part.data_ptr(),
                     self.mean_row.data_ptr(), self.scale.data_ptr(), self.lr_t.data_ptr(), i0, i1, t0, t1,
                     beta2t, self.eps[0], self.eps[1], self.clip_threshold, rel_step, self.weight_decay, grad_scale,
                     int(self.scale_parameter), self.counters.data_ptr(), has_matrix, 0)

    def rel_step(self, step: int) -> float:
        """adafactor.py:133-139"""
        if not self.relative_step:
            return float(self.lr)
        min_step = 1e-6 * step if self.warmup_init else 1e-2
        return min(min_step, 1.0 / math.sqrt(step))

    def refresh_param_norms(self) -> None:
        """Recompute the per-tile sums of p^2 (needed once, and again whenever the masters are changed from outside)."""
        a = self._args((0, self.ntensors, 0, self.nitems), 0.0, 0.0, 1.0)
        call("nk_adafactor_init", C.byref(a), ops._stream())
        self._p2_valid = True

    def step(self, grad_scale: float = 1.0) -> None:
        """One Adafactor update of every parameter (adafactor.py:162-255); also rewrites the bf16 shadows."""
        self.begin_step()
        if self.nstreams == 2 and self.store.master.is_cuda:
            cur = torch.cuda.current_stream()
            if self._side is None:
                self._side = [torch.cuda.Stream(device=self.store.master.device) for _ in range(2)]
            for st in self._side:
                st.wait_stream(cur)
            for ci in range(len(self.chunks)):
                with torch.cuda.stream(self._side[ci & 1]):
                    self.step_chunk(ci, grad_scale, ws_slot=ci & 1)
            for st in self._side:
                cur.wait_stream(st)
        else:
            for ci in range(len(self.chunks)):
                self.step_chunk(ci, grad_scale)
        self.end_step()

    # -- the same update, chunk by chunk: a chunk may be issued as soon as ITS gradients are final (DiffusionEngine streams the
    # update of each top-level block behind that block's backward).  Every per-step scalar (step count, beta2_t, relative step)
    # is fixed by begin_step(); the per-tensor statistics never cross a chunk, so the order of chunks does not matter. ------------
    def begin_step(self) -> None:
        if not self._p2_valid:
            self.refresh_param_norms()
        self.counters.zero_()
        self.step_count += 1
        self._beta2t = 1.0 - math.pow(self.step_count, self.decay_rate)
        self._rel = self.rel_step(self.step_count)
        self._done = [False] * len(self.chunks)

    def step_chunk(self, ci: int, grad_scale: float = 1.0, ws_slot: int = 0) -> None:
        if self._done[ci]:
            return
        if not self._mine(ci):          # another rank's shard
            self._done[ci] = True
            return
        a = self._args(self.chunks[ci], self._beta2t, self._rel, grad_scale, ws_slot)
        call("nk_adafactor_chunk", C.byref(a), ops._stream())
        self._done[ci] = True

    def chunks_in(self, tensor_lo: int, tensor_hi: int) -> list:
        """indices of the chunks that lie entirely inside tensors [tensor_lo, tensor_hi)"""
        return [ci for ci, c in enumerate(self.chunks) if c[0] >= tensor_lo and c[1] <= tensor_hi]

    def end_step(self, grad_scale: float = 1.0) -> None:
        for ci in range(len(self.chunks)):
            self.step_chunk(ci, grad_scale)
        self.store._mark_fresh()

    # -- introspection mirroring the reference's per-parameter state --------------------------------
    def current_lrs(self) -> Tensor:
        """Per-tensor lr of the last step (what `_get_lr` returned for each parameter)."""
        return self.lr_t

    def param_state(self, index: int) -> dict:
        t = self._tens_np[index]
        p = self.store.params[index]
        if t["kind"] == 1:
            return {"step": self.step_count, "exp_avg_sq_row": self.state[int(t["row_off"]):int(t["row_off"]) + int(t["d0"])],
                    "exp_avg_sq_col": self.state[int(t["col_off"]):int(t["col_off"]) + int(t["d1"])]}
        if t["kind"] == 2:
            O, I, KH, KW = p.shape
            row = self.state[int(t["row_off"]):int(t["row_off"]) + O * KH * I].view(O, KH, I).permute(0, 2, 1)
            col = self.state[int(t["col_off"]):int(t["col_off"]) + O * KW * I].view(O, KW, I).permute(0, 2, 1)
            return {"step": self.step_count, "exp_avg_sq_row": row, "exp_avg_sq_col": col}
        return {"step": self.step_count, "exp_avg_sq": self.state[int(t["row_off"]):int(t["row_off"]) + p.numel()].view(p.shape)}


    # -- checkpointing (torch.optim.Optimizer.state_dict layout, the reference class's keys: adafactor.py:186-204) ------
    def _rms(self, index: int) -> Tensor:
        p = self.store.params[index]
        return (ops._phys_flat(p).float().norm(2) / (p.numel() ** 0.5)).reshape(())

    def state_dict(self) -> dict:
        """{"state": {i: {...}}, "param_groups": [...]}: parameter i is `store.params[i]` (registration order, as
        torch numbers them).  Tensors are copies in the reference's logical shapes."""
        state = {}
        if self.step_count > 0:
            for i in range(self.ntensors):
                st = {k: (v.detach().clone().contiguous() if torch.is_tensor(v) else v) for k, v in self.param_state(i).items()}
                st["RMS"] = self._rms(i) if self.scale_parameter else 0
                state[i] = st
        group = dict(lr=self.lr, eps=self.eps, clip_threshold=self.clip_threshold, decay_rate=self.decay_rate, beta1=None,
                     weight_decay=self.weight_decay, scale_parameter=self.scale_parameter, relative_step=self.relative_step,
                     warmup_init=self.warmup_init, params=list(range(self.ntensors)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        state = sd.get("state", {})
        if not state:
            self.step_count = 0
            self.state.zero_()
            return
        steps = set()
        for i, st in state.items():
            i = int(i)
            if not 0 <= i < self.ntensors:
                raise ValueError(f"FlatAdafactor.load_state_dict: parameter index {i} out of range (0..{self.ntensors - 1})")
            mine = self.param_state(i)
            for k in ("exp_avg_sq_row", "exp_avg_sq_col", "exp_avg_sq"):
                if (k in mine) != (k in st):
                    raise ValueError(f"FlatAdafactor.load_state_dict: parameter {i} is {'not ' if k not in mine else ''}factored here but the checkpoint disagrees ({k})")
                if k in mine:
                    if tuple(mine[k].shape) != tuple(st[k].shape):
                        raise ValueError(f"FlatAdafactor.load_state_dict: {k} of parameter {i} has shape {tuple(st[k].shape)}, expected {tuple(mine[k].shape)}")
                    mine[k].copy_(st[k].to(mine[k].device, torch.float32))
            steps.add(int(st["step"]))
        if len(steps) != 1:
            raise ValueError(f"FlatAdafactor.load_state_dict: per-parameter step counts differ ({sorted(steps)}); the fused update keeps one")
        self.step_count = steps.pop()
        self._p2_valid = False       # RMS(p) is recomputed from the (separately restored) parameters


class AdafactorScheduler:
    """Proxy scheduler (adafactor.py:258-291): reports `initial_lr` before the first step, then the optimizer's own lr of
    the first parameter."""

    def __init__(self, optimizer: FlatAdafactor, initial_lr: float = 0.0):
        self.optimizer, self.initial_lr = optimizer, initial_lr

    def get_lr(self) -> list[float]:
        if self.optimizer.step_count == 0:
            return [self.initial_lr]
        return [float(self.optimizer.lr_t[0])]

    def step(self) -> None:
        pass


class FlatEma(torch.nn.Module):
    """`LitEma` (reference modules/ema.py:11-93) for a FlatParamStore: ONE flat fp32 shadow of every trainable parameter,
    updated by one kernel per step.  Same constructor arguments, decay warm-up `min(decay, (1+n)/(10+n))`, and the
    `store` / `copy_to` / `restore` protocol `DiffusionEngine.ema_scope` uses (models/diffusion.py:247-257).

    Checkpoint format = LitEma's: buffers `decay`, `num_updates` and one buffer per parameter named after it with the dots
    removed (ema.py:25-29); here those per-parameter entries are views of the flat shadow, written and read through
    `_save_to_state_dict` / `_load_from_state_dict`.  `names` are the parameter names as LitEma saw them (relative to the
    wrapped model, e.g. `diffusion_model.input_blocks.0.0.weight`); without names, positional `p<i>` keys are used."""

    def __init__(self, store, decay: float = 0.9999, use_num_updates: bool = True, names=None):
        super().__init__()
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.param_store = store
        self.register_buffer("decay", torch.tensor(decay, dtype=torch.float32))
        self.register_buffer("num_updates", torch.tensor(0 if use_num_updates else -1, dtype=torch.int))
        self._decay = float(decay)                       # host copies: update() must not read the device
        self._num_updates = 0 if use_num_updates else -1
        self.shadow = store.master.detach().clone()      # not a registered buffer: saved per parameter (below)
        self.collected: Optional[Tensor] = None
        names = list(names) if names is not None else [f"p{i}" for i in range(len(store.params))]
        if len(names) != len(store.params):
            raise ValueError("FlatEma: one name per store parameter expected")
        self.m_name2s_name = {n: n.replace(".", "") for n in names}
        self._s_names = [self.m_name2s_name[n] for n in names]
        store.add_listener(self)

    def masters_changed(self) -> None:
        """FlatParamStore hook.  Before the first update the average IS the model (LitEma clones the parameters at
        construction, after the checkpoint is loaded): follow the new weights.  Later the average is training state of its own."""
        if self._num_updates <= 0 and self.collected is None:
            self.shadow.copy_(self.param_store.master)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self.shadow = fn(self.shadow)
        return out

    def reset_num_updates(self) -> None:
        self._num_updates = 0
        self.num_updates.zero_()

    def update(self) -> None:
        """ema.py:40-59"""
        decay = self._decay
        if self._num_updates >= 0:
            self._num_updates += 1
            self.num_updates += 1
            decay = min(self._decay, (1 + self._num_updates) / (10 + self._num_updates))
        call("nk_ema_flat", self.shadow.data_ptr(), self.param_store.master.data_ptr(), self.shadow.numel(), float(1.0 - decay), ops._stream())

    def forward(self, model=None) -> None:   # LitEma is called as `self.model_ema(self.model)` (models/diffusion.py:244)
        self.update()

    def copy_to(self, model=None) -> None:
        """ema.py:61-68: the averaged weights become the model's (masters and bf16 shadows)."""
        self.param_store.master.copy_(self.shadow)
        self.param_store.refresh()

    def store(self, parameters=None) -> None:
        self.collected = self.param_store.master.detach().clone()

    def restore(self, parameters=None) -> None:
        if self.collected is None:
            raise RuntimeError("FlatEma.restore() without store()")
        self.param_store.master.copy_(self.collected)
        self.param_store.refresh()
        self.collected = None

    # -- LitEma-format checkpoint entries ---------------------------------------------------------------
    def _views(self):
        st = self.param_store
        return [(s, st._view(self.shadow, off, p)) for s, p, off in zip(self._s_names, st.params, st.offsets)]

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        for s_name, view in self._views():
            destination[prefix + s_name] = view if keep_vars else view.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        if prefix + "decay" in state_dict:
            self._decay = float(state_dict[prefix + "decay"])
        if prefix + "num_updates" in state_dict:
            self._num_updates = int(state_dict[prefix + "num_updates"])
        mine = set()
        for s_name, view in self._views():
            key = prefix + s_name
            mine.add(key)
            if key not in state_dict:
                if strict:
                    missing_keys.append(key)
                continue
            src = state_dict[key]
            if tuple(src.shape) != tuple(view.shape):
                error_msgs.append(f"size mismatch for {key}: checkpoint {tuple(src.shape)}, model {tuple(view.shape)}")
                continue
            with torch.no_grad():
                view.copy_(src)
        # the base class reports every key under the prefix that is not a registered buffer: the per-parameter entries are ours
        unexpected_keys[:] = [k for k in unexpected_keys if k not in mine]
