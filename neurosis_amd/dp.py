"""Data parallelism for the training step: a flat all-reduce of the UNet gradient buffer over RCCL/xGMI,
replacing Lightning's DDPStrategy (the reference has no collective call site of its own).

One process per GPU.  The UNet gradients already live in ONE contiguous fp32 buffer (FlatParamStore.grad) in
registration order, and the UNet's explicit backward finalises them from the back of that buffer to the front,
top-level block by block.  After each block the wrapper reduces that block's slice on a side stream, so the
exchange overlaps the rest of the backward; only the last slices (input blocks, time/label embeddings -- the
"backward tail") are exposed.  The mean over ranks is applied as grad_scale = 1/world in the fused optimizer.

Two exchange modes (NK_DP_MODE, or FlatDataParallel(mode=...)):

  allreduce (default)  every slice is all-reduced; every rank runs the whole optimizer.  What Lightning DDP does for the reference.
  rs_ag                the sharded form SURVEY 5 / 8(e) describes: the flat buffers are cut into `world` contiguous, TENSOR-ALIGNED shards
                       (factored Adafactor statistics never cross a shard); a slice's elements are REDUCED TO THEIR OWNER only
                       (reduce-scatter at tensor granularity), the owner runs the fused optimizer on its shard (1/world of the update:
                       11.4 -> ~1.4 ms per rank at 8 ranks for SDXL), and what the next forward reads is gathered (each rank broadcasts
                       its shard): the bf16 shadows, plus the fp32 masters of the parameters the kernels read in fp32 straight from the master
                       buffer -- biases, norm scales / shifts, the two channel-padded convolutions' weights -- packed into one small
                       buffer per shard (0.1 % of the elements).  fp32 bytes in + bf16 bytes out: 25 % fewer bytes per link than the all-reduce.  The fp32 MASTERS
                       of foreign shards' matrices go stale; `sync_masters()` gathers them where they are needed (checkpoints, EMA swaps).  Built and tested with gloo (world 2, CPU and two ranks on one GPU); not yet measured
                       on RCCL, so not the default.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn


class FlatGradReducer:
    """Sums slices [lo, hi) of a flat gradient tensor across ranks, asynchronously when the tensor is on a GPU.
    Device-agnostic so the exchange logic is testable with gloo on CPU."""

    def __init__(self, flat_grad: Tensor, group=None, wire_dtype: Optional[torch.dtype] = None, max_chunk: int = 1 << 28,
                 owner_bounds: Optional[list] = None):
        """owner_bounds: None = all-reduce every slice; else element offsets [0 = e_0 <= e_1 <= ... <= e_world = numel]: rank r owns
        [e_r, e_r+1) and a slice is reduced TO ITS OWNERS only (the other ranks' copies of it are left unspecified)."""
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.wire_dtype = wire_dtype
        self.max_chunk = max_chunk
        self.owner_bounds = owner_bounds
        self.collectives = 0           # collective calls issued since take_counts()
        self.wire_bytes = 0            # bytes each rank sends for them: all-reduce 2 (N-1)/N x payload, reduce / broadcast (N-1)/N x payload
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
        pieces = [(lo, hi, None)]
        if self.owner_bounds is not None:      # cut the slice at the shard boundaries: each piece goes to one owner
            pieces = []
            for r in range(self.world):
                a, b = max(lo, self.owner_bounds[r]), min(hi, self.owner_bounds[r + 1])
                if b > a:
                    pieces.append((a, b, r))
        for plo, phi, owner in pieces:
            for a in range(plo, phi, self.max_chunk):
                b = min(phi, a + self.max_chunk)
                sl = self.flat[a:b]
                if self.cuda:
                    with torch.cuda.stream(self.stream):
                        self._reduce(sl, owner)
                else:
                    self._reduce(sl, owner)
                self.reduced_elems += b - a

    def _reduce(self, sl: Tensor, owner: Optional[int] = None) -> None:
        wire = sl if self.wire_dtype is None or self.wire_dtype == sl.dtype else sl.to(self.wire_dtype)
        self.collectives += 1
        self.wire_bytes += int(wire.numel() * wire.element_size() * (2 if owner is None else 1) * (self.world - 1) / self.world)
        if owner is None or (wire.is_cuda and dist.get_backend(self.group) == "gloo"):
            # (gloo has no reduce for device tensors: the on-GPU rehearsal of the sharded mode all-reduces, which gives the owner the same sum)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
        else:
            dst = dist.get_global_rank(self.group, owner) if self.group is not None else owner
            dist.reduce(wire, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
        if wire is not sl and (owner is None or owner == self.rank):
            sl.copy_(wire)

    def take_counts(self):
        """(collective calls, bytes sent per rank) since the last call"""
        c = (self.collectives, self.wire_bytes)
        self.collectives = self.wire_bytes = 0
        return c

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

    def __init__(self, unet: nn.Module, store, group=None, wire_dtype: Optional[torch.dtype] = None, broadcast_params: bool = True,
                 mode: Optional[str] = None):
        import os

        self.unet, self.store, self.group = unet, store, group
        self.mode = mode or os.environ.get("NK_DP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"FlatDataParallel: mode must be 'allreduce' or 'rs_ag', got {self.mode!r}")
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.owner_bounds = self.tensor_bounds = None
        self._vec_index = None
        if self.mode == "rs_ag" and world > 1:
            self.tensor_bounds, self.owner_bounds = shard_bounds(store, world)
            # element indices of every shard's parameters that the forward reads as fp32 MASTERS, not through the bf16 shadows: the 1-D ones
            # (biases, norm scales / shifts) and the weights of channel-padded convolutions (nn.Conv2d.padded: the 4-channel latent's conv_in /
            # out, whose 8-channel stand-ins are rebuilt from the masters).  Gathered with the shadows.
            padded = {id(m.weight) for m in (unet.modules() if hasattr(unet, "modules") else ()) if getattr(m, "padded", False) and hasattr(m, "weight")}
            self._vec_index = []
            for r in range(world):
                spans = [torch.arange(store.offsets[t], store.offsets[t] + store.params[t].numel())
                         for t in range(self.tensor_bounds[r], self.tensor_bounds[r + 1])
                         if store.params[t] is not None and (store.params[t].dim() < 2 or id(store.params[t]) in padded)]
                idx = torch.cat(spans) if spans else torch.zeros(0, dtype=torch.long)
                self._vec_index.append(idx.to(store.master.device))
        self.reducer = FlatGradReducer(store.grad, group, wire_dtype, owner_bounds=self.owner_bounds)
        self.world = self.reducer.world
        self.rank = self.reducer.rank
        self.sync = True
        self._optimizers = []
        self._health = None
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
                self.reducer.reduce_range(lo, hi, also_wait=self.store.state.wgrad_stream)

    def no_sync(self, flag: bool = True) -> None:
        """Gradient accumulation: skip the exchange on all but the last micro-batch (DDP's no_sync)."""
        self.sync = not flag

    def finish(self) -> float:
        """Wait for the exchange; returns the grad_scale (1/world) the optimizer must apply for the mean.
        Also merges the device health word over the ranks (MAX), behind the last slice on the exchange stream: a rank whose stream-K fix-up
        gave up has poisoned its gradient tile with NaN, the reduction has spread it to every rank, and every rank's optimizer kernels must
        skip that update -- not only the flagged rank's."""
        self._merge_health()
        self.reducer.finish()
        return 1.0 / self.world

    def _merge_health(self) -> None:
        if self.world == 1 or not self.store.grad.is_cuda or not self.sync:
            return
        from .lib import call

        if self._health is None:
            self._health = torch.zeros(1, dtype=torch.int32, device=self.store.grad.device)
        main = torch.cuda.current_stream()
        ex = self.reducer.stream
        ex.wait_stream(main)                      # the backward (whose kernels may raise the word) is complete
        for s in (self.store.state.wgrad_stream,):
            if s is not None:
                ex.wait_stream(s)
        with torch.cuda.stream(ex):
            call("nk_health_export", self._health.data_ptr(), ex.cuda_stream)
            dist.all_reduce(self._health, op=dist.ReduceOp.MAX, group=self.group)
            call("nk_health_import", self._health.data_ptr(), ex.cuda_stream)

    # -- rs_ag: the optimizer runs on the owned shard only; shadows are gathered afterwards ------------------------------------------
    @property
    def sharded(self) -> bool:
        return self.owner_bounds is not None

    def owned_tensors(self) -> tuple[int, int]:
        """[lo, hi) indices into store.params of this rank's shard (all of them when not sharded)"""
        if not self.sharded:
            return (0, len(self.store.params))
        return (self.tensor_bounds[self.rank], self.tensor_bounds[self.rank + 1])

    def attach_optimizer(self, flat_optimizer) -> None:
        """Restrict a chunked flat optimizer (optim.FlatAdafactor) to this rank's shard.  No-op in all-reduce mode."""
        if self.sharded:
            flat_optimizer.restrict(*self.owned_tensors())
            self._optimizers.append(flat_optimizer)

    def after_optimizer_step(self) -> None:
        """rs_ag: every rank broadcasts the bf16 shadows of its shard (what the next forward reads), on the CURRENT stream -- call it
        where the optimizer kernels were issued (DiffusionEngine.optimizer_step does, on its optimizer stream)."""
        if not self.sharded:
            return
        for r in range(self.world):
            a, b = self.owner_bounds[r], self.owner_bounds[r + 1]
            if b > a:
                src = dist.get_global_rank(self.group, r) if self.group is not None else r
                dist.broadcast(self.store.shadow[a:b], src=src, group=self.group)
                self.reducer.collectives += 1
                self.reducer.wire_bytes += int((b - a) * self.store.shadow.element_size() * (self.world - 1) / self.world)
            idx = self._vec_index[r]
            if idx.numel():
                src = dist.get_global_rank(self.group, r) if self.group is not None else r
                pack = self.store.master[idx] if r == self.rank else torch.empty(idx.numel(), dtype=self.store.master.dtype, device=idx.device)
                dist.broadcast(pack, src=src, group=self.group)
                if r != self.rank:
                    self.store.master[idx] = pack
                self.reducer.collectives += 1
                self.reducer.wire_bytes += int(idx.numel() * 4 * (self.world - 1) / self.world)
        self.store._mark_fresh()

    def sync_masters(self) -> None:
        """rs_ag: gather the fp32 masters of every shard AND the attached optimizers' statistics for it (checkpoints, EMA swaps, anything
        that reads parameters other than through the bf16 shadows).  The shadows are already current."""
        if not self.sharded:
            return
        for r in range(self.world):
            a, b = self.owner_bounds[r], self.owner_bounds[r + 1]
            src = dist.get_global_rank(self.group, r) if self.group is not None else r
            if b > a:
                dist.broadcast(self.store.master[a:b], src=src, group=self.group)
            for o in self._optimizers:
                sa, sb = o.state_span(self.tensor_bounds[r], self.tensor_bounds[r + 1])
                if sb > sa:
                    dist.broadcast(o.state[sa:sb], src=src, group=self.group)
        for o in self.store.listeners:
            o.masters_changed()


def shard_bounds(store, world: int):
    """Cut store.params into `world` contiguous groups of about equal element count.  Returns (tensor index bounds, element offset
    bounds), each of length world + 1.  Shards begin at tensor boundaries: factored second moments and per-tensor RMS stay local."""
    offs = list(store.offsets) + [store.numel]
    total = store.numel
    tb = [0]
    for r in range(1, world):
        target = total * r // world
        # the tensor boundary nearest to the target, not before the previous bound
        best = min(range(tb[-1], len(store.params) + 1), key=lambda t: abs(offs[t] - target))
        tb.append(best)
    tb.append(len(store.params))
    return tb, [offs[t] for t in tb]
