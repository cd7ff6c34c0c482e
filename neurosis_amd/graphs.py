"""hipGraph replay of an explicit forward / backward closure chain.

The UNet is driven from Python: ~2 700 C-ABI launches per training step, ~28 us of host time each (closure bookkeeping, ctypes
marshalling, `hipLaunchKernel`), against kernels that take 2-100 us.  A launch sequence whose shapes repeat needs no host at
all: a round-2 micro-benchmark (`profiles/r02_micro_graph_cost.txt`) measured 0.3 us of host and 2.1 us of GPU time per dependent small kernel replayed from a
hipGraph on this ROCm, against 8.5 us launched one by one.

`ChainGraphs.run(fwd, inputs)` keeps, per input signature, one forward graph and a SEGMENTED backward over private memory pools:

* call 1 runs `fwd(*inputs)` eagerly on the capture stream (every lazily created per-stream resource of the C-ABI -- stream-K
  workspaces, function attributes -- exists afterwards);
* call 2 captures the forward (`fwd` on static input copies) into `g_f`, replays it, and hands back the chain's own output
  object plus a backward stub; the stub's first call captures `bwd(dout)` -- the closure that came out of the captured forward,
  so it reads the activations where `g_f` writes them;
* from then on a call is: copy the inputs into the static buffers, `g_f.replay()`; backward: copy `dout`, replay the segments.

Why the backward is not ONE graph.  The eager backward runs the dgrad chain on the main stream and every weight gradient on a
side stream; captured together they become two branches of one graph, and this ROCm's graph executor gives those branches almost
no overlap (198 ms/step against 185 eager on the metric's configuration; `DEBUG_HIP_FORCE_GRAPH_QUEUES` 1/2/3/8 all within 3 ms of
the fully serial 199.6).  So the capture cuts the chain where it already reports progress -- the end of each top-level block,
`EngineState.segment_hook` -- and keeps TWO single-stream graphs per segment: `M_k`, the main chain of block k, and `W_k`, the
weight-gradient launches of block k, which `ops.on_wgrad_stream` parks in `EngineState.deferred` during the capture instead of
issuing.  The replay launches M_k on the main stream and W_k on the side stream behind it, so W_k overlaps M_(k+1) exactly as the
eager streams do, at block granularity.  Between segments the replay calls the chain's gradient-ready hook (the data-parallel
exchange), as the eager chain does.

Memory safety of that arrangement: W_k reads tensors of the main pool (activations, output gradients) while M_(k+1) runs, so
nothing W_k reads may be recycled by a later M capture -- the parked closures, which hold those tensors, are kept alive until the
whole backward is captured; and W graphs allocate their workspaces from a pool of their own, which only the side stream's order
touches.  The price is the eager step's peak plus the gradient tensors that would otherwise have been recycled.

All signatures share the pools (a pair's activations are dead once its backward has run, and pairs never overlap), so a set of
aspect buckets costs the largest bucket's memory, not the sum.
"""
from __future__ import annotations

import gc
import os
from typing import Callable, Optional, Sequence

import torch
from torch import Tensor

from . import ops

__all__ = ["ChainGraphs", "ForwardGraphs", "frozen_stamp", "graphs_enabled"]


def graphs_enabled(kind: str = "unet") -> bool:
    """NK_GRAPH=0 keeps every chain eager, 1 graphs them all, a comma list names the kinds to graph ("unet", "vae", "te").  Read
    per call, so tools/ab_step.py can flip it.  Default "unet": the training chain's 2 700 launches are where the host time is
    (90 -> 15 ms per step); the frozen VAE encoder and text towers replayed from graphs measured 0.9 ms SLOWER each on the
    metric's configuration (184.9 / 185.8 / 185.75 / 186.5 ms for unet / +te / +vae / all, one box), so they stay eager unless asked."""
    v = os.environ.get("NK_GRAPH", "unet")
    if v in ("0", "1"):
        return v == "1"
    return kind in v.split(",")


class Stamps:
    """Diagnostic (tools/step_timeline.py): named device timestamps, nk_debug_stamp launches on the current stream -- captured into the
    replayed segments when enabled BEFORE the capture.  `read()` synchronises and returns {label: microseconds since the earliest}."""

    def __init__(self, device, slots: int = 1024):
        self.buf = torch.zeros(slots, dtype=torch.int64, device=device)
        self.slot: dict = {}

    def mark(self, label: str) -> None:
        from .lib import call

        i = self.slot.setdefault(label, len(self.slot))
        if i >= self.buf.numel():
            raise RuntimeError("neurosis_amd.graphs.Stamps: out of slots")
        call("nk_debug_stamp", self.buf.data_ptr() + 8 * i, ops._stream())

    def read(self) -> dict:
        torch.cuda.synchronize()
        v = self.buf.cpu().tolist()
        t0 = min(v[i] for i in self.slot.values() if v[i])
        return {k: (v[i] - t0) / 100.0 for k, i in self.slot.items() if v[i]}      # 100 MHz -> us


stamps: Optional[Stamps] = None      # set by the tool; None in production: no launch, no branch inside a captured sequence


def _mark(label: str) -> None:
    if stamps is not None:
        stamps.mark(label)


class _Pair:
    __slots__ = ("g_f", "segments", "static_in", "out", "closure", "dout", "dx", "gen", "warm", "bwd_replays", "tail", "cal", "sk_fp32")

    def __init__(self):
        self.g_f = None
        self.segments = None        # [(M_k, W_k or None, A_k or None, module_k or None)]
        self.static_in = None
        self.out = None
        self.closure = None
        self.dout = self.dx = None
        self.gen = 0
        self.warm = False
        self.bwd_replays = 0
        self.tail = None            # indices of the segments whose W_k replays on the MAIN stream behind the last M (see _replay_backward)
        self.cal = None             # events of the calibration replay
        self.sk_fp32 = False        # stream-K was eligible for fp32 weight gradients WHEN THE W GRAPHS WERE CAPTURED (no tail balancing then)


def _stream_k_may_take_fp32() -> bool:
    """NK_GEMM_SK as gemm.hip reads it (atoi on every call): 0 = never, 4 (default) = bf16 outputs only; 1 / 2 / 3 may hand an fp32 weight
    gradient to the stream-K kernel, whose per-stream workspace two concurrently replayed W graphs would share"""
    raw = os.environ.get("NK_GEMM_SK", "4").strip()
    try:
        v = int(raw)
    except ValueError:      # atoi semantics: leading digits, else 0
        digits = ""
        for ch in raw:
            if ch.isdigit() or (ch in "+-" and not digits):
                digits += ch
            else:
                break
        try:
            v = int(digits)
        except ValueError:
            v = 0
    return v not in (0, 4)


def _sig(t: Optional[Tensor]):
    return None if t is None else (tuple(t.shape), t.dtype, tuple(t.stride()))


class ChainGraphs:
    """Graphs for one chain (`fwd(*tensors) -> (out, bwd)`, `bwd(dout) -> dx or None`), keyed by input signature.
    `hook()` returns the chain's current gradient-ready hook (or None)."""

    def __init__(self, owner: Tensor, hook: Optional[Callable] = None):
        self.owner = owner              # any parameter of the chain: names the engine (side stream, accumulate flag)
        self.hook = hook or (lambda: None)
        self.pairs = {}
        self.pool = self.pool_w = self.pool_a = None
        self.stream: Optional[torch.cuda.Stream] = None
        self.replays = 0                # graph launches so far (tests)
        self.generation = 0             # forward replays / captures so far, over ALL signatures: they share one activation pool
        self.broken = False             # a capture failed: the chain stays eager (still HIP kernels, launched from Python)

    def clear(self) -> None:
        self.pairs.clear()

    # ------------------------------------------------------------------------------------------------
    def run(self, fwd: Callable, inputs: Sequence[Optional[Tensor]], extra_key=()):
        mode = ops.wgrad_mode(self.owner)
        key = (extra_key, tuple(_sig(t) for t in inputs), mode, ops.recording())     # (a forward that keeps activations for a backward is another launch sequence)
        pair = self.pairs.get(key)
        if pair is None:
            pair = self.pairs[key] = _Pair()
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.owner.device)
            self.pool, self.pool_w = (torch.cuda.graph_pool_handle() for _ in range(2))
            self.ticks = torch.zeros(1, dtype=torch.int32, device=self.owner.device)   # backward replays so far, counted on the device
        if not pair.warm or self.broken:
            return self._warm(pair, fwd, inputs)
        if pair.g_f is None:
            try:
                self._capture_forward(pair, fwd, inputs)
            except Exception as err:      # a runtime that cannot capture this chain: say so, once, and keep launching from Python
                import warnings

                warnings.warn(f"neurosis_amd.graphs: hipGraph capture of the forward chain failed ({type(err).__name__}: {err}); "
                              "this chain stays on the eager launch path (NK_GRAPH=0 silences the attempt)")
                self.broken = True
                torch.cuda.synchronize()
                return self._warm(pair, fwd, inputs)
        else:
            for s, t in zip(pair.static_in, inputs):
                if t is not None:
                    s.copy_(t, non_blocking=True)
            pair.g_f.replay()
            self.replays += 1
        pair.gen += 1
        self.generation += 1
        gen, chain_gen = pair.gen, self.generation

        def bwd(dout: Tensor):
            # (the signatures of a chain share ONE activation pool: a forward of ANY of them since this one has overwritten what this backward reads)
            if gen != pair.gen or chain_gen != self.generation:
                raise RuntimeError("neurosis_amd.graphs: backward of a forward that has since been replayed (its activations were overwritten)")
            # the overwrite / add mode of the weight-gradient kernels is a launch argument baked into the captured backward: it is part of
            # the signature, taken at forward time -- a caller that flips it between forward and backward would replay the wrong mode for ever
            if ops.wgrad_mode(self.owner) != mode:
                raise RuntimeError("neurosis_amd.graphs: the gradient accumulate mode changed between this chain's forward and its backward; "
                                   "set it (DiffusionEngine.accumulate) BEFORE the forward of the micro-batch")
            if pair.segments is None:
                self._capture_backward(pair, dout)
            else:
                pair.dout.copy_(dout, non_blocking=True)
            self._replay_backward(pair)
            return pair.dx

        return pair.out, bwd

    # ------------------------------------------------------------------------------------------------
    def _warm(self, pair: _Pair, fwd: Callable, inputs):
        """The first pass of a signature: eager, but on the capture stream, so that whatever the C-ABI creates lazily per
        stream exists before a capture (which may not allocate device memory behind torch's back) needs it."""
        main = torch.cuda.current_stream()
        cs = self.stream
        cs.wait_stream(main)
        with torch.cuda.stream(cs):
            out, closure = fwd(*inputs)
        main.wait_stream(cs)
        for t in inputs:
            if t is not None:
                t.record_stream(cs)
        pair.warm = True

        def bwd(dout: Tensor):
            cs.wait_stream(torch.cuda.current_stream())
            dout.record_stream(cs)
            with torch.cuda.stream(cs):
                dx = closure(dout)
                ops.join_wgrad_stream(self.owner)
            torch.cuda.current_stream().wait_stream(cs)
            return dx

        return out, bwd

    @staticmethod
    def _quiesce() -> None:
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()

    def _capture_forward(self, pair: _Pair, fwd: Callable, inputs) -> None:
        pair.static_in = [None if t is None else t.clone() for t in inputs]
        self._quiesce()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream), ops.capture_scope():
            g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
            try:
                _mark("F.begin")
                pair.out, pair.closure = fwd(*pair.static_in)
                _mark("F.end")
            finally:
                g.capture_end()
        pair.g_f = g
        g.replay()
        self.replays += 1

    def _capture_backward(self, pair: _Pair, dout: Tensor) -> None:
        st = ops.state_of(self.owner)
        side = st.wgrad_stream
        pair.dout = dout.clone()
        closure, pair.closure = pair.closure, None
        segments, held = [], []
        cur = [None]
        self._quiesce()

        def begin():
            cur[0] = torch.cuda.CUDAGraph()
            cur[0].capture_begin(pool=self.pool, capture_error_mode="thread_local")
            _mark(f"M{len(segments)}.begin")

        def cut(module=None):
            """End of a segment of the main chain: close M_k, capture what the segment parked for the side stream as W_k."""
            if module is None:
                self.ticks.add_(1)        # (the tail after the last block may hold no launch at all: an empty graph cannot be instantiated)
            _mark(f"M{len(segments)}.end")
            cur[0].capture_end()
            g_m, g_w = cur[0], None
            parked, st.deferred = st.deferred, ([] if side is not None else None)
            if parked:
                g_w = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    # a pool of its OWN: the W graphs behind the end of the main chain replay on two streams at once (_replay_backward),
                    # and graphs that share a pool may share the memory of their temporaries (split-K / partial-sum workspaces)
                    g_w.capture_begin(pool=torch.cuda.graph_pool_handle(), capture_error_mode="thread_local")
                    try:
                        _mark(f"W{len(segments)}.begin")
                        for fn, _reads in parked:
                            fn()
                        _mark(f"W{len(segments)}.end")
                    finally:
                        g_w.capture_end()
                held.append(parked)       # the closures hold what W_k reads: nothing of it may be recycled by a later M capture
            segments.append((g_m, g_w, module))

        def boundary(module):
            cut(module)
            begin()

        st.deferred = [] if side is not None else None
        st.segment_hook = boundary
        try:
            with torch.cuda.stream(self.stream), ops.capture_scope():
                begin()
                try:
                    pair.dx = closure(pair.dout)
                finally:
                    cut(None)
        finally:
            st.deferred = None
            st.segment_hook = None
        del closure
        held.clear()
        pair.segments = segments
        pair.sk_fp32 = _stream_k_may_take_fp32()      # the workspace is baked into the graphs now: what the variable says later is irrelevant

    def _replay_backward(self, pair: _Pair) -> None:
        """M_0, W_0 | M_1, W_1 | ...: M_k on the main stream, W_k on the side stream behind it.  The main chain ends before the side stream does
        (the last blocks' weight gradients: 2.9 ms of exposed tail at SDXL batch 4, profiles/r04_step_timeline.txt): single-GPU, the W graphs
        that START behind the last M are shared out between BOTH streams (longest first).  Which ones those are is measured once, by events
        around the third replay of a signature; the kernels and their arguments are the same either way."""
        st = ops.state_of(self.owner)
        side = st.wgrad_stream
        main = torch.cuda.current_stream()
        hook = self.hook()
        # (the stream-K workspace -- flags, tile counter, partial tiles -- is keyed by the launch stream AT CAPTURE TIME and baked into the W
        # graphs: two of them replayed at once would share it.  The default NK_GEMM_SK=4 never gives an fp32 weight gradient to stream-K; under
        # the A/B settings 1 / 2 / 3 it may, so those run without tail balancing: ADVICE round 4)
        # (decided by what held at CAPTURE time -- pair.sk_fp32 -- not by the live variable: tools flip NK_GEMM_SK in-process, and
        # graphs captured under 1 / 2 / 3 must never be balanced after it goes back to 4)
        balance = side is not None and hook is None and os.environ.get("NK_TAIL_BALANCE", "1") != "0" and not pair.sk_fp32
        pair.bwd_replays += 1
        if balance and pair.tail is None and pair.cal is not None and pair.cal["done"].query():
            pair.tail = self._plan_tail(pair.cal)
            pair.cal = None
        calibrate = balance and pair.tail is None and pair.cal is None and pair.bwd_replays == 3
        cal = None
        if calibrate:
            ev = lambda: torch.cuda.Event(enable_timing=True)
            cal = {"t0": ev(), "m": [], "w": {}, "done": ev()}
            cal["t0"].record(main)
        tail = pair.tail if balance and pair.tail else ()
        late = []
        for k, (g_m, g_w, module) in enumerate(pair.segments):
            g_m.replay()
            if cal is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record(main)
                cal["m"].append(e)
            if g_w is not None:
                if k in tail:
                    late.append(g_w)
                else:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        if cal is not None:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(side)
                            g_w.replay()
                            e1.record(side)
                            cal["w"][k] = (e0, e1)
                        else:
                            g_w.replay()
            if hook is not None and module is not None:
                hook(module)
            self.replays += 1
        for g_w in late:                  # behind every M on the main stream; each W_k only needs its own M_k
            g_w.replay()
        if side is not None:
            main.wait_stream(side)
        if cal is not None:
            cal["done"].record(main)
            pair.cal = cal

    @staticmethod
    def _plan_tail(cal) -> frozenset:
        """From one timed replay: the W graphs that began after the main chain had ended, split over two bins by longest-processing-time;
        the side stream's bin starts with what was still running there when the main chain ended."""
        t0 = cal["t0"]
        main_end = t0.elapsed_time(cal["m"][-1])
        late, side_busy = [], 0.0
        for k, (e0, e1) in cal["w"].items():
            start, end = t0.elapsed_time(e0), t0.elapsed_time(e1)
            if start >= main_end - 0.02:
                late.append((end - start, k))
            elif end > main_end:
                side_busy = max(side_busy, end - main_end)
        load = {"side": side_busy, "main": 0.0}
        on_main = []
        for dur, k in sorted(late, reverse=True):
            if dur < 0.05:                  # (a handful of tiny launches: not worth a second queue)
                continue
            b = "main" if load["main"] < load["side"] else "side"
            load[b] += dur
            if b == "main":
                on_main.append(k)
        return frozenset(on_main)


def _tree_map(fn: Callable, obj):
    if isinstance(obj, Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _tree_map(fn, v) for k, v in obj.items()}
    if isinstance(obj, (tuple, list)):
        return type(obj)(_tree_map(fn, v) for v in obj)
    return obj


def frozen_stamp(module: torch.nn.Module):
    """What a captured launch sequence over `module`'s weights is valid for.  Kernels read bf16 shadows by ADDRESS: a
    store-managed parameter's shadow is a view of the store's flat buffer, rewritten in place by the fused optimizers and by
    refresh() (stable: the store's identity is enough); any other parameter's shadow is a cached cast keyed on the tensor's
    version and address (ops.shadow), re-created when a checkpoint load or `.to()` changes either."""
    out = []
    for p in module.parameters():
        st = getattr(p, "_nk_store", None)
        out.append(("store", id(st), st.shadow.data_ptr()) if st is not None else (p._version, p.data_ptr()))
    return hash(tuple(out))


class ForwardGraphs:
    """hipGraph replay of a forward-only launch sequence (the frozen VAE encoder, the frozen text towers): `run(fn, inputs)` with
    `fn(*tensors) -> tensor | tuple | dict of tensors`.  First call of a signature: eager on the capture stream; second: capture;
    then replay.  The outputs live in the graph's pool and are overwritten by the next replay, so every call hands out clones
    (they are a few MB: latents, text embeddings)."""

    def __init__(self, device):
        self.device = device
        self.entries = {}
        self.pool = None
        self.stream: Optional[torch.cuda.Stream] = None
        self.replays = 0

    def run(self, fn: Callable, inputs: Sequence[Optional[Tensor]], extra_key=()):
        key = (extra_key, tuple(_sig(t) for t in inputs))
        e = self.entries.get(key)
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
            self.pool = torch.cuda.graph_pool_handle()
        if e is None:                                     # warm-up pass, eager, on the capture stream (see ChainGraphs._warm)
            self.entries[key] = {"g": None}
            main = torch.cuda.current_stream()
            self.stream.wait_stream(main)
            with torch.cuda.stream(self.stream):
                out = fn(*inputs)
            main.wait_stream(self.stream)
            for t in inputs:
                if t is not None:
                    t.record_stream(self.stream)
            _tree_map(lambda t: t.record_stream(main), out)
            return out
        if e["g"] is None:
            e["in"] = [None if t is None else t.clone() for t in inputs]
            ChainGraphs._quiesce()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(self.stream), ops.capture_scope():
                g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
                try:
                    e["out"] = fn(*e["in"])
                finally:
                    g.capture_end()
            e["g"] = g
        else:
            for s, t in zip(e["in"], inputs):
                if t is not None:
                    s.copy_(t, non_blocking=True)
        e["g"].replay()
        self.replays += 1
        return _tree_map(torch.clone, e["out"])
