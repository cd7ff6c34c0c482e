"""Data parallelism for the training step: a flat all-reduce of the UNet gradient buffer over RCCL/xGMI,
replacing Lightning's DDPStrategy (the reference has no collective call site of its own).

One process per GPU.  The UNet gradients already live in ONE contiguous fp32 buffer (FlatParamStore.grad) in
registration order, and the UNet's explicit backward finalises them from the back of that buffer to the front,
top-level block by block.  After each block the wrapper reduces that block's slice on a side stream, so the
exchange overlaps the rest of the backward; only the last slices (input blocks, time/label embeddings -- the
"backward tail") are exposed.  The mean over ranks is applied as grad_scale = 1/world in the fused optimizer.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn


class FlatGradReducer:
    """Sums slices [lo, hi) of a flat gradient tensor across ranks, asynchronously when the tensor is on a GPU.
    Device-agnostic so the exchange logic is testable with gloo on CPU."""

    def __init__(self, flat_grad: Tensor, group=None, wire_dtype: Optional[torch.dtype] = None, max_chunk: int = 1 << 28):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.wire_dtype = wire_dtype
        self.max_chunk = max_chunk
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending = []
        self.reduced_elems = 0
        # optional instrumentation (bench.py): events on the exchange stream around the first / last collective of a step
        self.record_timing = False
        self.ev_first = None
        self.ev_last = None

    def reduce_range(self, lo: int, hi: int, also_wait=None) -> None:
        """Reduce flat[lo:hi] once everything enqueued so far on the current stream -- and on `also_wait` (the
        weight-gradient stream that wrote part of the slice) -- has finished.  Only the EXCHANGE stream waits: the compute
        stream is not held up."""
        if self.world == 1 or hi <= lo:
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            for other in (also_wait if isinstance(also_wait, (list, tuple)) else [also_wait]):
                if other is not None:
                    self.stream.wait_stream(other)
            if self.record_timing and self.ev_first is None:
                self.ev_first = torch.cuda.Event(enable_timing=True)
                self.ev_first.record(self.stream)
        for a in range(lo, hi, self.max_chunk):
            b = min(hi, a + self.max_chunk)
            sl = self.flat[a:b]
            if self.cuda:
                with torch.cuda.stream(self.stream):
                    self._reduce(sl)
            else:
                self._reduce(sl)
            self.reduced_elems += b - a

    def _reduce(self, sl: Tensor) -> None:
        if self.wire_dtype is not None and self.wire_dtype != sl.dtype:
            tmp = sl.to(self.wire_dtype)
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            sl.copy_(tmp)
        else:
            dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group)

    def finish(self) -> None:
        """Make the compute stream wait for every outstanding reduction."""
        if self.cuda and self.world > 1:
            ev = torch.cuda.Event(enable_timing=self.record_timing)
            ev.record(self.stream)
            torch.cuda.current_stream().wait_event(ev)
            if self.record_timing:
                self.ev_last = ev
        self.reduced_elems = 0

    def take_timing(self):
        """(first-collective-start, last-collective-end) events of the step just finished, then reset."""
        pair = (self.ev_first, self.ev_last)
        self.ev_first = self.ev_last = None
        return pair


class FlatDataParallel:
    """Wires a FlatGradReducer to a UNetModel's backward through `grad_ready_hook`."""

    def __init__(self, unet: nn.Module, store, group=None, wire_dtype: Optional[torch.dtype] = None, broadcast_params: bool = True):
        self.unet, self.store = unet, store
        self.reducer = FlatGradReducer(store.grad, group, wire_dtype)
        self.world = self.reducer.world
        self.sync = True
        if broadcast_params and self.world > 1:
            dist.broadcast(store.master, src=0, group=group)
            store.refresh()
        unet.grad_ready_hook = self._on_block_done

    def _on_block_done(self, module: nn.Module) -> None:
        if self.sync:
            lo, hi = self.store.param_range(module)
            from . import ops

            import os

            # NK_DP_JOIN=1: the compute stream itself waits for the weight-gradient stream before the slice is handed over.
            # Only useful with a host-synchronous backend (the gloo rehearsal blocks the host inside all_reduce until the
            # exchange stream's dependencies are done: 14.8 vs 30 s/step); with RCCL the collective is stream-ordered.
            if os.environ.get("NK_DP_JOIN") == "1":
                ops.join_wgrad_stream(self.store.params[0])
                self.reducer.reduce_range(lo, hi)
            else:
                # (under graph replay the small parameter-gradient reductions of the slice run on a third stream)
                self.reducer.reduce_range(lo, hi, also_wait=[self.store.state.wgrad_stream, self.store.state.aux_stream])

    def no_sync(self, flag: bool = True) -> None:
        """Gradient accumulation: skip the exchange on all but the last micro-batch (DDP's no_sync)."""
        self.sync = not flag

    def finish(self) -> float:
        """Wait for the exchange; returns the grad_scale (1/world) the optimizer must apply for the mean."""
        self.reducer.finish()
        return 1.0 / self.world
