"""Data parallelism for the training step: a flat all-reduce of the UNet gradient buffer over RCCL/xGMI,
replacing Lightning's DDPStrategy (the reference has no collective call site of its own).

One process per GPU.  The UNet gradients already live in ONE contiguous fp32 buffer (FlatParamStore.grad) in
registration order, and the UNet's explicit backward finalises them from the back of that buffer to the front,
top-level block by block.  After each block the wrapper reduces that block's slice on a side stream, so the
exchange overlaps the rest of the backward; only the last slices (input blocks, time/label embeddings -- the
"backward tail") are exposed.  The mean over ranks is applied as grad_scale = 1/world in the fused optimizer.

Two exchange modes (NK_DP_MODE, or FlatDataParallel(mode=...)):

  allreduce (default)  every slice is all-reduced; every rank runs the whole optimizer.  What Lightning DDP does for the reference.
  rs_ag                the sharded form SURVEY 5 / 8(e) describes, built from the two collectives a point-to-point xGMI mesh runs at full
                       width: EVERY slice (top-level block) of the flat buffers is cut into `world` TENSOR-ALIGNED parts of about equal size
                       (factored Adafactor statistics never cross a part) and rank r owns part r of every slice.  A slice's gradients go
                       through `reduce_scatter_tensor` (its parts copied into equal-size padded rows of a staging buffer, the rank's row of
                       the sum copied back), the rank runs the fused optimizer on its parts (1/world of the update and of its 14 B/parameter
                       of HBM traffic), and what the next forward reads comes back through `all_gather_into_tensor`: the bf16 shadows slice by
                       slice, plus ONE small gather of the fp32 masters that kernels read directly (biases, norm scales / shifts, the two
                       channel-padded convolutions' weights: 0.1 % of the elements).  fp32 in + bf16 out: 25 % fewer bytes per link than the
                       all-reduce.  (Round 3 reduced each part to its owner with rooted `reduce` / `broadcast`: on RCCL a rooted collective
                       is a chain through one root and leaves six of the seven links idle.)  The fp32 MASTERS of foreign parts' matrices go
                       stale; `sync_masters()` gathers them where they are needed (checkpoints, EMA swaps; DiffusionEngine.state_dict calls
                       it).  Tested with gloo (world 2 and 3 on CPU, two ranks on one GPU); never run on RCCL -- no multi-GPU box was
                       available to this build -- so not the default.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn


def _forced() -> bool:
    import os

    return os.environ.get("NK_DP_FORCE", "0") == "1"


class FlatGradReducer:
    """All-reduces slices [lo, hi) of a flat gradient tensor across ranks, asynchronously when the tensor is on a GPU; also owns the
    exchange stream, the counters and the timing events that the sharded mode (FlatDataParallel, rs_ag) shares.
    Device-agnostic so the exchange logic is testable with gloo on CPU."""

    def __init__(self, flat_grad: Tensor, group=None, wire_dtype: Optional[torch.dtype] = None, max_chunk: int = 1 << 28):
        self.flat = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # a single rank has nothing to exchange; NK_DP_FORCE=1 issues the collectives all the same (tests/test_dp_gpu.py: the one way a 1-GPU box
        # can run the exchange through RCCL itself -- communicator, streams, collective argument checks -- rather than through gloo)
        self.active = self.world > 1 or (dist.is_initialized() and _forced())
        self.wire_dtype = wire_dtype
        self.max_chunk = max_chunk
        self.collectives = 0           # collective calls issued since take_counts()
        self.wire_bytes = 0            # bytes each rank sends for them: all-reduce 2 (N-1)/N x payload, reduce-scatter / all-gather (N-1)/N x payload
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending = []
        self.reduced_elems = 0
        # optional instrumentation (bench.py): events on the exchange stream around the first / last collective of a step
        self.record_timing = False
        self.ev_first = None
        self.ev_last = None

    def order_after_compute(self, also_wait=None) -> None:
        """The EXCHANGE stream waits for everything enqueued so far on the current stream -- and on `also_wait` (the weight-gradient stream
        that wrote part of the slice); the compute stream is not held up."""
        if not self.cuda:
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.stream.wait_event(ev)
        for other in (also_wait if isinstance(also_wait, (list, tuple)) else [also_wait]):
            if other is not None:
                self.stream.wait_stream(other)
        if self.record_timing and self.ev_first is None:
            self.ev_first = torch.cuda.Event(enable_timing=True)
            self.ev_first.record(self.stream)

    def reduce_range(self, lo: int, hi: int, also_wait=None) -> None:
        """All-reduce flat[lo:hi] once the gradients in it are final (order_after_compute)."""
        if not self.active or hi <= lo:
            return
        self.order_after_compute(also_wait)
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
        wire = sl if self.wire_dtype is None or self.wire_dtype == sl.dtype else sl.to(self.wire_dtype)
        self.collectives += 1
        self.wire_bytes += int(wire.numel() * wire.element_size() * 2 * (self.world - 1) / self.world)
        dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
        if wire is not sl:
            sl.copy_(wire)

    def take_counts(self):
        """(collective calls, bytes sent per rank) since the last call"""
        c = (self.collectives, self.wire_bytes)
        self.collectives = self.wire_bytes = 0
        return c

    def finish(self) -> None:
        """Make the compute stream wait for every outstanding reduction."""
        if self.cuda and self.active:
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


class SlicePlan:
    """One slice (top-level block) of the flat buffers cut into `world` tensor-aligned parts: part r = elements [cuts[r], cuts[r + 1]) =
    tensors [tcuts[r], tcuts[r + 1]).  `row` = elements per padded staging row (the largest part, rounded up to 64)."""

    def __init__(self, store, lo: int, hi: int, world: int):
        offs = list(store.offsets) + [store.numel]
        ts = [t for t in range(len(store.params)) if lo <= offs[t] < hi]
        if not ts or offs[ts[0]] != lo or ts != list(range(ts[0], ts[-1] + 1)):
            raise ValueError(f"rs_ag: the slice [{lo}, {hi}) does not begin at a tensor / is not a run of tensors")
        t0, t1 = ts[0], ts[-1] + 1
        ends = offs[t0:t1] + [hi]              # candidate cut points: tensor starts, and the end of the slice
        spans = [ends[i + 1] - ends[i] for i in range(t1 - t0)]

        def fill(cap):                         # greedy runs of whole tensors, each <= cap elements: the cut indices, or None if > world runs
            cuts_i, acc = [0], 0
            for i, n in enumerate(spans):
                if acc and acc + n > cap:
                    cuts_i.append(i)
                    acc = 0
                acc += n
            return cuts_i + [len(spans)] if len(cuts_i) <= world else None

        # the smallest row (= largest part) any split into <= world runs of whole tensors can have: the staging rows are padded to it
        lo_cap, hi_cap = max(spans), hi - lo
        while lo_cap < hi_cap:
            mid = (lo_cap + hi_cap) // 2
            if fill(mid) is not None:
                hi_cap = mid
            else:
                lo_cap = mid + 1
        idx = fill(lo_cap)
        # ranks past the last run own nothing of this slice.  That only happens for slices with fewer tensors than ranks -- SDXL's time_embed,
        # label_emb, input_blocks.0 and out heads: 14 tensors, 10 M of 2 567 M elements -- so the low ranks' extra optimizer work is < 0.4 %;
        # every other block has hundreds of tensors and min-max parts.  (Rotating the first owner per slice would cost the equal-parts fast
        # path of _reduce_scatter / after_optimizer_step, which needs part r in row r.)
        idx += [len(spans)] * (world + 1 - len(idx))
        self.lo, self.hi = lo, hi
        self.tcuts = [t0 + i for i in idx]
        self.cuts = [ends[i] for i in idx]
        self.sizes = [self.cuts[r + 1] - self.cuts[r] for r in range(world)]
        self.row = (max(self.sizes) + 63) // 64 * 64


class FlatDataParallel:
    """Wires the exchange of the flat gradient buffer to a UNetModel's backward through `grad_ready_hook`."""

    def __init__(self, unet: nn.Module, store, group=None, wire_dtype: Optional[torch.dtype] = None, broadcast_params: bool = True,
                 mode: Optional[str] = None, slices: Optional[list] = None, max_chunk: Optional[int] = None):
        """slices (rs_ag): the element ranges [lo, hi) the backward hands over, one per top-level block, tiling the flat buffers; default:
        the UNet's own top-level blocks.  max_chunk: elements per collective call (NK_DP_MAX_CHUNK, default 2^28)."""
        import os

        self.unet, self.store, self.group = unet, store, group
        self.mode = mode or os.environ.get("NK_DP_MODE", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"FlatDataParallel: mode must be 'allreduce' or 'rs_ag', got {self.mode!r}")
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        max_chunk = int(max_chunk or os.environ.get("NK_DP_MAX_CHUNK", 1 << 28))
        self.plans = None                   # rs_ag: {(lo, hi): SlicePlan}
        self._vec_index = None
        self._stage = {}
        if self.mode == "rs_ag" and (world > 1 or (dist.is_initialized() and _forced())):
            if wire_dtype is not None and wire_dtype != store.grad.dtype:
                raise ValueError("rs_ag reduces in the gradient buffer's dtype (a narrower wire is an all-reduce option)")
            ranges = sorted(slices if slices is not None else _top_block_ranges(unet, store))
            # a block's range ends at its last parameter's last element; the alignment padding up to the next block's first tensor rides with it
            nexts = [b[0] for b in ranges[1:]] + [store.numel]
            if not ranges or ranges[0][0] != 0 or any(a[1] > n for a, n in zip(ranges, nexts)):
                raise ValueError("rs_ag: the slices must tile the flat buffers")
            self.plans = {(lo, hi): SlicePlan(store, lo, n, world) for (lo, hi), n in zip(ranges, nexts)}
            # element indices of every rank's parameters that the forward reads as fp32 MASTERS, not through the bf16 shadows: the 1-D ones
            # (biases, norm scales / shifts) and the weights of channel-padded convolutions (nn.Conv2d.padded: the 4-channel latent's conv_in /
            # out, whose 8-channel stand-ins are rebuilt from the masters).  Gathered with the shadows, in one collective.
            padded = {id(m.weight) for m in (unet.modules() if hasattr(unet, "modules") else ()) if getattr(m, "padded", False) and hasattr(m, "weight")}
            self._vec_index = []
            for r in range(world):
                spans = [torch.arange(store.offsets[t], store.offsets[t] + store.params[t].numel())
                         for pl in self.plans.values() for t in range(pl.tcuts[r], pl.tcuts[r + 1])
                         if store.params[t] is not None and (store.params[t].dim() < 2 or id(store.params[t]) in padded)]
                idx = torch.cat(spans) if spans else torch.zeros(0, dtype=torch.long)
                self._vec_index.append(idx.to(store.master.device))
        self.reducer = FlatGradReducer(store.grad, group, wire_dtype, max_chunk=max_chunk)
        self.world = self.reducer.world
        self.rank = self.reducer.rank
        self.sync = True
        self.masters_whole = True           # rs_ag: False from the first sharded update until sync_masters() has run on every rank
        self._optimizers = []
        self._health = None
        if broadcast_params and self.reducer.active:
            dist.broadcast(store.master, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            store.refresh()
        unet.grad_ready_hook = self._on_block_done
        try:
            store.dp = self            # DiffusionEngine.state_dict() asks it to make the masters whole (rs_ag)
        except AttributeError:
            pass

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
                self.exchange_range(lo, hi)
            else:
                self.exchange_range(lo, hi, also_wait=self.store.state.wgrad_stream)

    def exchange_range(self, lo: int, hi: int, also_wait=None) -> None:
        """Hand the finished gradients flat[lo:hi] to the exchange: all-reduced (default mode), or reduce-scattered to the parts' owners."""
        if not self.sharded:
            self.reducer.reduce_range(lo, hi, also_wait=also_wait)
            return
        if hi <= lo:
            return
        plan = self.plans.get((lo, hi))
        if plan is None:
            raise ValueError(f"rs_ag: [{lo}, {hi}) is not one of the planned slices")
        red = self.reducer
        red.order_after_compute(also_wait)
        if red.cuda:
            with torch.cuda.stream(red.stream):
                self._reduce_scatter(plan)
        else:
            self._reduce_scatter(plan)

    # -- rs_ag ------------------------------------------------------------------------------------------------------------------------
    def _staging(self, name: str, n: int, dtype) -> Tensor:
        buf = self._stage.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            # the slice staging rows (rs_* / ag_*) are sized once for the largest slice; the master-vector gather (vec_*) is ~0.1 % of the
            # elements and is sized by what it holds (it used to inherit the > 1 GB slice size)
            grow = n
            if name.startswith(("rs_", "ag_")):
                grow = max(n, max((pl.row for pl in self.plans.values()), default=0) * (self.world if name.endswith("_all") else 1))
            buf = self._stage[name] = torch.empty(grow, dtype=dtype, device=self.store.grad.device)
        return buf[:n]

    def _gloo_on_device(self, t: Tensor) -> bool:
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _reduce_scatter(self, plan: SlicePlan) -> None:
        """grad[plan.cuts[me] : plan.cuts[me + 1]] <- sum over ranks; the other parts of the slice are left as they were (their owners have them)"""
        g, W, me, row = self.store.grad, self.world, self.rank, plan.row
        if all(sz == row for sz in plan.sizes):               # equal, aligned parts: the slice IS the staging layout
            rows = g[plan.lo:plan.hi]
        else:
            rows = self._staging("rs_all", W * row, g.dtype)
            for r in range(W):
                if plan.sizes[r]:
                    rows[r * row:r * row + plan.sizes[r]].copy_(g[plan.cuts[r]:plan.cuts[r + 1]])
        mine = self._staging("rs_mine", row, g.dtype)
        if self._gloo_on_device(g):        # (gloo has no reduce-scatter for device tensors: the on-GPU rehearsal all-reduces the staging rows)
            dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group)
            mine.copy_(rows[me * row:(me + 1) * row])
            self.reducer.wire_bytes += int(2 * rows.numel() * rows.element_size() * (W - 1) / W)
        else:
            dist.reduce_scatter_tensor(mine, rows, op=dist.ReduceOp.SUM, group=self.group)
            self.reducer.wire_bytes += int(rows.numel() * rows.element_size() * (W - 1) / W)
        self.reducer.collectives += 1
        if plan.sizes[me]:
            g[plan.cuts[me]:plan.cuts[me + 1]].copy_(mine[:plan.sizes[me]])

    def _all_gather(self, mine: Tensor, rows: Tensor) -> None:
        if self._gloo_on_device(mine):
            parts = list(rows.view(self.world, -1).unbind(0))
            dist.all_gather(parts, mine, group=self.group)
        else:
            dist.all_gather_into_tensor(rows, mine, group=self.group)
        self.reducer.collectives += 1
        self.reducer.wire_bytes += int(rows.numel() * rows.element_size() * (self.world - 1) / self.world)

    @property
    def sharded(self) -> bool:
        return self.plans is not None

    def owned_ranges(self, rank: Optional[int] = None) -> list:
        """[(tensor_lo, tensor_hi), ...] of store.params that `rank` (default: this one) updates -- one range per slice; everything when
        not sharded"""
        r = self.rank if rank is None else rank
        if not self.sharded:
            return [(0, len(self.store.params))]
        return [(pl.tcuts[r], pl.tcuts[r + 1]) for pl in self.plans.values() if pl.tcuts[r + 1] > pl.tcuts[r]]

    def owns(self, tensor_index: int, rank: Optional[int] = None) -> bool:
        return any(a <= tensor_index < b for a, b in self.owned_ranges(rank))

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
        if not self.reducer.active or not self.store.grad.is_cuda or not self.sync:
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

    def attach_optimizer(self, flat_optimizer) -> None:
        """Restrict a chunked flat optimizer (optim.FlatAdafactor) to this rank's parts.  No-op in all-reduce mode."""
        if self.sharded:
            flat_optimizer.restrict_ranges(self.owned_ranges())
            self._optimizers.append(flat_optimizer)

    def after_optimizer_step(self) -> None:
        """rs_ag: gather what the next forward reads -- every slice's bf16 shadows, then the fp32 masters that kernels read directly -- on
        the CURRENT stream: call it where the optimizer kernels were issued (DiffusionEngine.optimizer_step does, on its optimizer stream)."""
        if not self.sharded:
            return
        self.masters_whole = self.world == 1
        sh, W, me = self.store.shadow, self.world, self.rank
        for plan in self.plans.values():
            row = plan.row
            equal = all(sz == row for sz in plan.sizes)
            rows = sh[plan.lo:plan.hi] if equal else self._staging("ag_all", W * row, sh.dtype)
            mine = rows[me * row:(me + 1) * row] if equal else self._staging("ag_mine", row, sh.dtype)
            if not equal and plan.sizes[me]:
                mine[:plan.sizes[me]].copy_(sh[plan.cuts[me]:plan.cuts[me + 1]])
            self._all_gather(mine.clone() if equal else mine, rows)       # (in place is not allowed: the input must not alias the output)
            if not equal:
                for r in range(W):
                    if r != me and plan.sizes[r]:
                        sh[plan.cuts[r]:plan.cuts[r + 1]].copy_(rows[r * row:r * row + plan.sizes[r]])
        vmax = (max(int(i.numel()) for i in self._vec_index) + 63) // 64 * 64
        if vmax:
            m = self.store.master
            rows = self._staging("vec_all", W * vmax, m.dtype)
            mine = self._staging("vec_mine", vmax, m.dtype)
            idx = self._vec_index[me]
            if idx.numel():
                mine[:idx.numel()].copy_(m[idx])
            self._all_gather(mine, rows)
            for r in range(W):
                idx = self._vec_index[r]
                if r != me and idx.numel():
                    m[idx] = rows[r * vmax:r * vmax + idx.numel()]
        self.store._mark_fresh()

    def sync_masters(self) -> None:
        """rs_ag: gather the fp32 masters of every part AND the attached optimizers' statistics for it (checkpoints, EMA swaps, anything
        that reads parameters other than through the bf16 shadows).  The shadows are already current.  A collective: every rank calls it.
        (Rooted broadcasts: a checkpoint-time path, not a per-step one.)"""
        if not self.sharded:
            return
        for plan in self.plans.values():
            for r in range(self.world):
                src = dist.get_global_rank(self.group, r) if self.group is not None else r
                if plan.sizes[r]:
                    dist.broadcast(self.store.master[plan.cuts[r]:plan.cuts[r + 1]], src=src, group=self.group)
                for o in self._optimizers:
                    sa, sb = o.state_span(plan.tcuts[r], plan.tcuts[r + 1])
                    if sb > sa:
                        dist.broadcast(o.state[sa:sb], src=src, group=self.group)
        for o in self.store.listeners:
            o.masters_changed()
        self.masters_whole = True


def _top_block_ranges(unet, store) -> list:
    """element ranges of the UNet's top-level blocks, in the order the backward finalises them reversed (= registration order): what
    grad_ready_hook reports, one slice each"""
    tops = [getattr(unet, "time_embed", None), getattr(unet, "label_emb", None), *getattr(unet, "input_blocks", ()), getattr(unet, "middle_block", None),
            *getattr(unet, "output_blocks", ()), getattr(unet, "out", None)]
    out = []
    for m in tops:
        if m is None:
            continue
        lo, hi = store.param_range(m)
        if hi > lo:
            out.append((lo, hi))
    return out
