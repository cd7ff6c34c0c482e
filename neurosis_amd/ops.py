"""Tensor-level wrappers over the C-ABI (neurosis_amd.lib) and the forward/backward op pairs the modules
are composed from.

Conventions
  * activations are bf16 CUDA(=HIP) tensors.  A "token matrix" is a 2-D tensor [M, C] whose last stride is
    1 (it may be a column slice of a wider buffer).  An image batch is an `Img`: a token matrix of its
    pixels in channels-last order plus (N, H, W).
  * every `*_fwd` returns `(out, bwd)`; `bwd(grad_out, ...)` returns the input gradient(s) and writes the
    parameter gradients IN PLACE into `param.grad` (fp32), through the HIP kernels.  Nothing here calls
    a PyTorch compute kernel on activation-sized data.
  * PyTorch provides memory (torch.empty), streams and autograd plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from dataclasses import dataclass
from typing import Callable, Optional

import torch
from torch import Tensor

from . import lib
from .lib import NkAttnDesc, NkConvDesc, call, query

BF16 = torch.bfloat16


class EngineState:
    """Training-step flags of ONE engine (= one FlatParamStore; `store.state`).  They used to be process-global, which let two
    engines in one process (the GAN step's autoencoder and discriminator stores, a test next to a model) leak the accumulate
    flag and the side stream into each other.  Parameters that belong to no store use the default instance `ops.state`."""

    _live = weakref.WeakSet()

    def __init__(self, wgrad_stream=None):
        self.grad_accumulate = False  # False: weight-grad kernels overwrite; True: they add (micro-batch accumulation)
        self.assume_zeroed = False  # True: a caller guarantees .grad buffers are all-zero before the first micro-batch (split-K skips its memset)
        self.wgrad_stream = wgrad_stream  # optional side HIP stream: weight-gradient GEMMs run there, concurrently with the dgrad chain
        # the small same-shape weight gradients of one transformer block (attn1.to_out, attn2.to_q, attn2.to_out: 1280 x 1280 =
        # 80 tiles of 128 x 160 each) as ONE batched launch (nk_linear_wgrad_batched): 240 tiles = one full round of the two-group
        # kernel, 49.7 us for the three against 3 x 45.4 us one by one (tools/bench_g2.py).  (Round 1 measured batching -2 % with the
        # 128 x 128 kernels, which gained nothing from the fuller grid, and with the block's LARGE weight gradients deferred too.)
        # In the real two-stream step the step TIME does not move (tools/ab_step.py, interleaved rounds, GC frozen: 184.4 vs 185.3,
        # 188.0 vs 187.8, 188.4 vs 188.0 ms; bench.py with graph replay 186.4 / 188.8 vs 186.9 / 187.1): the side stream is not the
        # critical path.  What moves is the kernel time behind it: 140 fewer launches per step, linear weight gradients 500 -> 570
        # TFLOP/s serialized (314 -> 339 in the step), the tile engine 644 -> 662 serialized and 498 -> 508 in the step.  On by default since the chain
        # replays from hipGraphs (the deferred launches used to lengthen the host path); NK_BATCH_WGRADS=0 turns it off.
        self.batch_wgrads = os.environ.get("NK_BATCH_WGRADS", "1") != "0"
        # LayerNorm gamma / beta gradients on the weight-gradient stream.  Launched from Python this measured SLOWER (217.6 vs
        # 203.9 ms/step in round 1, +5.5 ms in round 2: 210 more cross-stream event waits per step delay the weight-gradient GEMMs
        # queued behind them).  Under hipGraph replay those waits are gone -- the parked launches become part of each segment's
        # side-stream graph -- and the main chain sheds ~4 ms of small kernels: 177.2 / 177.1 vs 178.2 / 178.7 ms/step (bench.py,
        # alternating).  So: None = automatic -- on exactly for launches that are being CAPTURED for replay (`deferred` is a list), off for
        # every eager launch (NK_GRAPH=0, a chain whose capture failed, the frozen towers, bench.py's instrumented eager replays); True / False force it.
        self.norm_params_on_side_stream: Optional[bool] = None
        # hipGraph capture of a backward chain (neurosis_amd/graphs.py): while `deferred` is a list, on_wgrad_stream() parks the
        # side-stream work there instead of launching it, and the chain reports the end of each top-level block to
        # `segment_hook`, which captures the parked launches as that segment's own graph
        self.deferred: Optional[list] = None
        self.segment_hook: Optional[Callable] = None
        EngineState._live.add(self)

    def derived(self) -> "EngineState":
        """A state that shares this one's side stream but always overwrites (channel-padded stand-in parameters)."""
        return EngineState(self.wgrad_stream)


_capture_depth = 0


def capturing() -> bool:
    """True while neurosis_amd.graphs records a launch sequence: caches of weight-derived tensors must be refilled (in place)
    rather than trusted, so that the refill is part of the graph."""
    return _capture_depth > 0


class capture_scope:
    def __enter__(self):
        global _capture_depth
        _capture_depth += 1

    def __exit__(self, *exc):
        global _capture_depth
        _capture_depth -= 1


state = EngineState()
state.param_epoch = 0  # process-wide "some parameter changed" counter (keys of captured graphs); bumped by every store


def state_of(p) -> EngineState:
    """The engine state that governs parameter `p`: its own tag, its store's, or the default."""
    if p is not None:
        st = getattr(p, "_nk_state", None)
        if st is not None:
            return st
        store = getattr(p, "_nk_store", None)
        if store is not None:
            return store.state
    return state


def wgrad_mode(p=None) -> int:
    """accumulate argument of the weight-gradient kernels (see include/neurosis_hip.h) for parameter `p`."""
    st = state_of(p)
    if st.grad_accumulate:
        return 1
    return 2 if st.assume_zeroed else 0


# Whether the forward now running will be followed by a backward (its closures are kept).  torch.is_grad_enabled() cannot answer that
# inside the autograd bridge: torch.autograd.Function.forward always runs with grad mode off.  nn.NkFunction.forward raises the flag around
# `run`; the activation-dropping policies (ResBlock.use_checkpoint, BasicTransformerBlock.checkpoint / .recompute) read it.
# The flag is per THREAD: a validation or data-loader thread that runs a forward of its own must not see the training thread's.
# Callers that drive `module.fwd(...)` and the returned closure themselves (outside nn.NkFunction) and want the checkpoint / recompute
# policies honoured wrap the forward in `with ops.recording_backward():` -- without it the blocks keep their activations (correct, not lean).
_recording = threading.local()


class recording_backward:
    """`with ops.recording_backward():` -- the forward launched inside will be followed by a backward through its closure."""

    def __enter__(self):
        _recording.depth = getattr(_recording, "depth", 0) + 1

    def __exit__(self, *exc):
        _recording.depth -= 1


def recording() -> bool:
    return getattr(_recording, "depth", 0) > 0


def on_wgrad_stream(fn: Callable[[], None], *reads: Tensor, owner=None) -> None:
    """Run `fn` (kernels that only WRITE parameter gradients of `owner`'s engine) on that engine's side stream if it has one.

    Weight-gradient GEMMs are off the critical path of backward (nothing downstream reads them until the optimizer /
    all-reduce), and most of them -- like the dgrad GEMMs they sit next to -- do not fill 256 CUs x 2 workgroups on
    their own (e.g. 1280x1280 outputs = 100 tiles).  Issuing them on a second HIP stream lets the hardware co-schedule
    both kernels' workgroups, which recovers the tile-quantisation tail of each without split-K atomics.
    `reads` are the activation tensors fn consumes: they are pinned to the side stream for the caching allocator."""
    st = state_of(owner)
    side = st.wgrad_stream
    if side is None:
        fn()
        return
    if st.deferred is not None:
        st.deferred.append((fn, reads))
        return
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    for t in reads:
        if t is not None:
            t.record_stream(side)


class WgradQueue:
    """Collects weight-gradient GEMMs and issues those of identical shape as ONE batched launch (blockIdx.z).
    Used per transformer block: its attn1.to_out / attn2.to_q / attn2.to_out gradients are three 100-tile grids."""

    def __init__(self, owner=None):
        self.items = []  # (dy, x, dw2d)
        self.colparts = []  # (part, dgamma, dbeta, nrows, C, accumulate)
        self.owner = owner

    MAX_ELEMS = 2 << 20     # weights up to 1280 x 1280: larger ones fill the chip alone and should not wait for the end of the block

    @classmethod
    def takes(cls, weight: Tensor) -> bool:
        return weight.numel() <= cls.MAX_ELEMS

    def add(self, dy: Tensor, x: Tensor, dw: Tensor, db: Optional[Tensor] = None) -> None:
        """db: the layer's bias gradient [N] (fp32), summed from dy by the same launch"""
        self.items.append((dy, x, dw, db))

    def add_colpart(self, part: Tensor, dgamma: Tensor, dbeta: Tensor, nrows: int, Cc: int, accumulate: bool) -> None:
        """partial rows of a one-pass LayerNorm backward (nk_layernorm_bwd_rows): reduced with the block's other ones in one launch"""
        self.colparts.append((part, dgamma, dbeta, nrows, Cc, accumulate))

    def flush(self) -> None:
        colparts, self.colparts = self.colparts, []
        if colparts:
            on_wgrad_stream(lambda: colpart_reduce(colparts), *[c[0] for c in colparts], owner=self.owner)
        items, self.items = self.items, []
        if not items:
            return
        groups = {}
        for it in items:
            dy, x, dw, _db = it
            key = (dy.shape[0], dy.shape[1], x.shape[1], dy.stride(0), x.stride(0), dw.stride(0))
            groups.setdefault(key, []).append(it)
        mode = wgrad_mode(self.owner)

        def run():
            for (M, N, K, lddy, ldx, lddw), its in groups.items():
                for i in range(0, len(its), 8):
                    chunk = its[i:i + 8]
                    if len(chunk) == 1:
                        gemm_tn_f32(chunk[0][0], chunk[0][1], chunk[0][2], mode, dbias=chunk[0][3])
                        continue
                    n = len(chunk)
                    arr = C.c_void_p * n
                    call("nk_linear_wgrad_batched", arr(*[c[0].data_ptr() for c in chunk]), arr(*[c[1].data_ptr() for c in chunk]),
                         arr(*[c[2].data_ptr() for c in chunk]), arr(*[None if c[3] is None else c[3].data_ptr() for c in chunk]), n, M, N, K,
                         lddy, ldx, lddw, mode, _stream())

        on_wgrad_stream(run, *[t for it in items for t in it[:2]], owner=self.owner)


def colpart_reduce(entries) -> None:
    """dgamma / dbeta of several one-pass LayerNorm backwards from their partial rows, NK_COLPART_MAX per launch (current stream)"""
    for i in range(0, len(entries), lib.NK_COLPART_MAX):
        chunk = entries[i:i + lib.NK_COLPART_MAX]
        b = lib.NkColpartBatch()
        for z, (part, dgamma, dbeta, nrows, Cc, acc) in enumerate(chunk):
            b.part[z], b.dgamma[z], b.dbeta[z] = part.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
            b.nrows[z], b.C[z], b.accumulate[z] = nrows, Cc, int(acc)
        b.n = len(chunk)
        call("nk_colpart_reduce_batch", C.byref(b), _stream())


_wgrad_queue: Optional[WgradQueue] = None


class batched_wgrads:
    """Context manager: inside it, linear_fwd's backward defers bias-free bookkeeping of weight gradients to a queue that is
    flushed (batched by shape) on exit."""

    def __init__(self, owner=None):
        self.owner = owner      # a parameter of the block: selects the engine whose flags apply

    def __enter__(self):
        global _wgrad_queue
        self.prev = _wgrad_queue
        _wgrad_queue = WgradQueue(self.owner) if state_of(self.owner).batch_wgrads else None
        return _wgrad_queue

    def __exit__(self, *exc):
        global _wgrad_queue
        q, _wgrad_queue = _wgrad_queue, self.prev
        if exc[0] is None and q is not None:
            q.flush()
        return False


def join_wgrad_stream(owner=None) -> None:
    """Make the current stream wait for every weight-gradient kernel issued so far (on `owner`'s engine's side stream, or, with
    no owner, on every live engine's)."""
    if owner is not None:
        st = state_of(owner)
        side = st.wgrad_stream
        if side is not None and st.deferred is None:
            torch.cuda.current_stream().wait_stream(side)
        return
    seen = set()
    for st in list(EngineState._live):
        side = st.wgrad_stream
        if side is not None and st.deferred is None and id(side) not in seen:
            seen.add(id(side))
            torch.cuda.current_stream().wait_stream(side)


def _ws(n_floats: int, device) -> Tensor:
    return torch.empty(n_floats, dtype=torch.float32, device=device)


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _check2d(t: Tensor, name: str):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != BF16 or not t.is_cuda:
        raise ValueError(f"{name}: expected a 2-D bf16 CUDA tensor with unit inner stride, got {tuple(t.shape)} {t.dtype} {t.stride()} {t.device}")


@dataclass
class Img:
    """Channels-last image batch: t is [N*H*W, C] (row = pixel)."""

    t: Tensor
    N: int
    H: int
    W: int
    sums: Optional[Tensor] = None    # GroupNorm sums of THIS tensor, [N, 2G] fp32 (entry 2g = sum, 2g+1 = sum of squares per group), when the
    #                                  kernel that produced it emitted them (conv2d_fwd(stats_groups=G)): groupnorm_fwd then skips its statistics pass

    @property
    def C(self) -> int:
        return self.t.shape[1]

    @staticmethod
    def from_nchw(x: Tensor) -> "Img":
        """View a logical-NCHW bf16 tensor in channels_last memory format as an Img (no copy); otherwise convert."""
        N, Cc, H, W = x.shape
        if x.dtype == BF16 and x.permute(0, 2, 3, 1).is_contiguous():
            return Img(x.permute(0, 2, 3, 1).reshape(N * H * W, Cc), N, H, W)
        return Img(nchw_to_tokens(x, Cc), N, H, W)

    def to_nchw(self) -> Tensor:
        """Logical NCHW view (channels_last memory format) of the same storage."""
        return self.t.view(self.N, self.H, self.W, self.C).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------------
# parameter plumbing: bf16 shadows and fp32 grads
# ------------------------------------------------------------------------------------------------
def _param_stamp(p: Tensor):
    """What a cached derivative of parameter `p` (bf16 shadow, channel-padded copy) is valid for.  Store-managed parameters are
    updated by raw-pointer kernels, which torch's version counter does not see: they go by the global epoch the store bumps.
    Everything else (frozen VAE / text-encoder weights, stand-alone modules) goes by the tensor's own version counter and
    address, so that an optimizer step on the UNet does not invalidate 850 M frozen parameters' shadows every step."""
    st = getattr(p, "_nk_store", None)
    if st is not None:
        return ("epoch", id(st), st.epoch)
    return ("version", p._version, p.data_ptr())


def shadow(p: Tensor) -> Tensor:
    """bf16 copy of an fp32 parameter in the same physical layout (flat store view, or cached cast)."""
    if getattr(p, "_nk_store", None) is not None:
        return p._nk_shadow            # store-managed: rewritten with the masters by the fused optimizers / store.refresh()
    s = getattr(p, "_nk_shadow", None)
    stamp = _param_stamp(p)
    if s is not None and getattr(p, "_nk_shadow_stamp", None) == stamp:
        return s
    flat = _phys_flat(p)
    n = flat.numel()
    if s is not None and n % 8 == 0 and s.numel() == n and s.device == p.device:
        # same tensor, new values (an in-place update): re-cast into the SAME buffer -- captured graphs hold its address
        call("nk_cast_f32_to_bf16", flat.data_ptr(), s.data_ptr(), n, _stream())
        p._nk_shadow_stamp = stamp
        return s
    s = torch.empty(flat.numel(), dtype=BF16, device=p.device)
    if n % 8 == 0:
        call("nk_cast_f32_to_bf16", flat.data_ptr(), s.data_ptr(), n, _stream())
    else:  # tiny odd-sized parameters
        pad = torch.zeros((n + 7) // 8 * 8, dtype=torch.float32, device=p.device)
        pad[:n] = flat
        s8 = torch.empty(pad.numel(), dtype=BF16, device=p.device)
        call("nk_cast_f32_to_bf16", pad.data_ptr(), s8.data_ptr(), pad.numel(), _stream())
        s = s8[:n]
    p._nk_shadow = s
    p._nk_shadow_stamp = stamp
    return s


def _phys_flat(p: Tensor) -> Tensor:
    """The parameter's storage as a flat fp32 tensor in physical order (conv weights: [O][KH][KW][I])."""
    d = p.detach()
    if d.dim() == 4:
        d = d.permute(0, 2, 3, 1)
    if not d.is_contiguous():
        raise ValueError("parameter storage must be dense (conv weights in channels_last memory format)")
    return d.reshape(-1)


def grad_flat(p: Tensor) -> Tensor:
    """fp32 gradient buffer of p in physical order (allocated zeroed on first use)."""
    if p.grad is None:
        g = torch.zeros_like(p, memory_format=torch.preserve_format)
        p.grad = g
    g = p.grad
    if g.dim() == 4:
        g = g.permute(0, 2, 3, 1)
    if not g.is_contiguous():
        raise ValueError("parameter .grad must share the parameter's physical layout")
    return g.reshape(-1)


def w2d(p: Tensor) -> Tensor:
    """bf16 shadow as the [out, K] matrix the kernels read."""
    return shadow(p).view(p.shape[0], -1)


def g2d(p: Tensor) -> Tensor:
    return grad_flat(p).view(p.shape[0], -1)


def conv_weight_param(out_ch: int, in_ch: int, kh: int, kw: int) -> torch.nn.Parameter:
    """fp32 OIHW parameter whose storage is [O][KH][KW][I] (channels_last), the layout the kernels read."""
    base = torch.empty(out_ch, kh, kw, in_ch)
    return torch.nn.Parameter(base.permute(0, 3, 1, 2))


# ------------------------------------------------------------------------------------------------
# layout / dtype boundary
# ------------------------------------------------------------------------------------------------
def nchw_to_tokens(x: Tensor, cpad: int, scale: float = 1.0) -> Tensor:
    """NCHW (fp32 or bf16, contiguous) -> channels-last bf16 [N*H*W, cpad] (extra channels zero)."""
    N, Cc, H, W = x.shape
    x = x.contiguous()
    if x.dtype not in (torch.float32, BF16):
        raise ValueError(f"unsupported dtype {x.dtype}")
    out = torch.empty(N * H * W, cpad, dtype=BF16, device=x.device)
    call("nk_nchw_to_nhwc", x.data_ptr(), int(x.dtype == torch.float32), out.data_ptr(), N, Cc, H * W, cpad, scale, _stream())
    return out


def tokens_to_nchw(t: Tensor, N: int, Cc: int, H: int, W: int, dtype=torch.float32) -> Tensor:
    """channels-last bf16 [N*H*W, cpad] -> contiguous NCHW tensor of `dtype` holding the first Cc channels."""
    _check2d(t, "t")
    out = torch.empty(N, Cc, H, W, dtype=dtype, device=t.device)
    call("nk_nhwc_to_nchw", t.data_ptr(), out.data_ptr(), int(dtype == torch.float32), N, Cc, H * W, t.shape[1], _stream())
    return out


def cast_bf16(x: Tensor) -> Tensor:
    """fp32 -> bf16 of a contiguous tensor (numel % 8 == 0)."""
    if x.dtype == BF16:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    n = x.numel()
    if n % 8:
        raise ValueError("cast_bf16 needs numel % 8 == 0")
    call("nk_cast_f32_to_bf16", x.data_ptr(), out.data_ptr(), n, _stream())
    return out


def add(a: Tensor, b: Tensor) -> Tensor:
    if a.shape != b.shape or not a.is_contiguous() or not b.is_contiguous():
        raise ValueError("add: contiguous tensors of equal shape expected")
    out = torch.empty_like(a)
    call("nk_add", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# Linear
# ------------------------------------------------------------------------------------------------
def gemm_nt(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, alpha: float = 1.0, out: Optional[Tensor] = None) -> Tensor:
    """y = alpha * x @ w^T + bias + residual ; x [M,K], w [N,K] bf16, bias fp32 [N], residual bf16 [M,N]."""
    _check2d(x, "x")
    _check2d(w, "w")
    M, K = x.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"gemm_nt: K mismatch {x.shape} vs {w.shape}")
    if out is None:
        out = torch.empty(M, N, dtype=BF16, device=x.device)
    if residual is not None:
        _check2d(residual, "residual")
    call("nk_linear_fwd", x.data_ptr(), w.data_ptr(), _p(bias), _p(residual), out.data_ptr(), M, N, K, x.stride(0), w.stride(0),
         residual.stride(0) if residual is not None else 0, out.stride(0), float(alpha), _stream())
    return out


def gemm_nt_batched(xs, ws, outs=None):
    """[x_i @ w_i^T] for same-shape bias-free problems, up to eight per launch (nk_linear_fwd_batched).  xs / ws: lists of 2-D bf16
    matrices (unit inner stride, equal shapes and row strides); returns the list of outputs."""
    M, K = xs[0].shape
    N = ws[0].shape[0]
    for x, w in zip(xs, ws):
        _check2d(x, "x")
        _check2d(w, "w")
        if x.shape != (M, K) or w.shape != (N, K) or x.stride(0) != xs[0].stride(0) or w.stride(0) != ws[0].stride(0):
            raise ValueError("gemm_nt_batched: every problem must have the same shape and strides")
    if outs is None:
        outs = [torch.empty(M, N, dtype=BF16, device=xs[0].device) for _ in xs]
    for i in range(0, len(xs), 8):
        n = min(8, len(xs) - i)
        arr = C.c_void_p * n
        call("nk_linear_fwd_batched", arr(*[t.data_ptr() for t in xs[i:i + n]]), arr(*[t.data_ptr() for t in ws[i:i + n]]),
             arr(*[t.data_ptr() for t in outs[i:i + n]]), n, M, N, K, xs[0].stride(0), ws[0].stride(0), outs[0].stride(0), _stream())
    return outs


def gemm_nn(dy: Tensor, w: Tensor, dx_add: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """dx = dy @ w + dx_add ; dy [M,N], w [N,K]."""
    _check2d(dy, "dy")
    _check2d(w, "w")
    M, N = dy.shape
    K = w.shape[1]
    if w.shape[0] != N:
        raise ValueError(f"gemm_nn: N mismatch {dy.shape} vs {w.shape}")
    if out is None:
        out = torch.empty(M, K, dtype=BF16, device=dy.device)
    if dx_add is not None:
        _check2d(dx_add, "dx_add")
    call("nk_linear_dgrad", dy.data_ptr(), w.data_ptr(), _p(dx_add), out.data_ptr(), M, N, K, dy.stride(0), w.stride(0),
         dx_add.stride(0) if dx_add is not None else 0, out.stride(0), _stream())
    return out


def gemm_tn_f32(dy: Tensor, x: Tensor, dw: Tensor, accumulate, dbias: Optional[Tensor] = None) -> None:
    """dw (+)= dy^T @ x ; dy [M,N], x [M,K], dw fp32 [N,K].  dbias fp32 [N] (+)= column sums of dy, from the same launch."""
    _check2d(dy, "dy")
    _check2d(x, "x")
    M, N = dy.shape
    K = x.shape[1]
    if x.shape[0] != M or dw.shape != (N, K) or dw.dtype != torch.float32 or dw.stride(1) != 1:
        raise ValueError(f"gemm_tn_f32: bad shapes {dy.shape} {x.shape} {dw.shape}")
    if dbias is None:
        call("nk_linear_wgrad", dy.data_ptr(), x.data_ptr(), dw.data_ptr(), M, N, K, dy.stride(0), x.stride(0), dw.stride(0), int(accumulate), _stream())
        return
    if dbias.shape != (N,) or dbias.dtype != torch.float32 or not dbias.is_contiguous():
        raise ValueError(f"gemm_tn_f32: dbias must be a dense fp32 [{N}] tensor")
    call("nk_linear_wgrad_bias", dy.data_ptr(), x.data_ptr(), dw.data_ptr(), dbias.data_ptr(), M, N, K, dy.stride(0), x.stride(0), dw.stride(0),
         int(accumulate), _stream())


def colsum(dy: Tensor, out: Tensor, accumulate: bool) -> None:
    _check2d(dy, "dy")
    M, N = dy.shape
    ws = _ws(query("nk_colsum_ws_floats", M, N), dy.device)
    call("nk_colsum", dy.data_ptr(), out.data_ptr(), ws.data_ptr(), M, N, dy.stride(0), int(accumulate), _stream())


def linear_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Optional[Tensor] = None, need_dx: bool = True, x_saved=None):
    """nn.Linear forward on a token matrix, with optional fused residual add.
    bwd(dy, dx_add=None) -> dx (None if need_dx is False); writes weight.grad / bias.grad.
    x_saved: a zero-argument callable that REBUILDS x for the weight gradient (selective recompute: the caller drops x after the forward)."""
    y = gemm_nt(x, w2d(weight), bias, residual)
    return y, _linear_bwd(x if x_saved is None else x_saved, weight, bias, need_dx)


def geglu_save_enabled() -> bool:
    """NK_GEGLU_SAVE: 1 (default) = FeedForward keeps s = [gelu(g) | a gelu'(g)] of its projection instead of u = [a | g] (round 6: the GEGLU
    derivative is evaluated where the cdf already is -- the forward's epilogue -- and the fused input gradient multiplies); 0 = u (A/B runs)"""
    return os.environ.get("NK_GEGLU_SAVE", "1") != "0"


def linear_geglu_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], x_saved=None, save_derivative: bool = False):
    """FeedForward.net[0] = GEGLU (modules/attention.py:50-57): u = x @ weight^T + bias [M, 2I] and h = u[:, :I] * gelu(u[:, I:]) [M, I].
    One launch (the GEGLU in the projection's epilogue, nk_linear_fwd_geglu) where the 256 x 256 kernel takes the shape -- the two SDXL
    FeedForward widths at batch 4 -- else the GEMM followed by the GEGLU kernel.  Returns (u, h, bwd); bwd(du) as linear_fwd's.
    save_derivative: the first result is s = [gelu(g) | a gelu'(g)] [M, 2I] instead of u (u is never written): hand it to the output
    projection's backward as `geglu_s` (nk_linear_dgrad_geglu_s)."""
    _check2d(x, "x")
    M, K = x.shape
    I2 = weight.shape[0]
    I = I2 // 2
    wq = w2d(weight)
    if I2 % 2 == 0 and x.is_contiguous() and query("nk_linear_fwd_geglu_ok", M, I, K):
        u = torch.empty(M, I2, dtype=BF16, device=x.device)
        h = torch.empty(M, I, dtype=BF16, device=x.device)
        call("nk_linear_fwd_geglu_s" if save_derivative else "nk_linear_fwd_geglu", x.data_ptr(), wq.data_ptr(), _p(bias), u.data_ptr(), h.data_ptr(), M, I, K,
             x.stride(0), wq.stride(0), u.stride(0), h.stride(0), _stream())
    else:
        u = gemm_nt(x, wq, bias, None)
        if save_derivative:
            h = torch.empty(M, I, dtype=BF16, device=x.device)
            call("nk_geglu_fwd_s", u.data_ptr(), h.data_ptr(), u.data_ptr(), M, I, _stream())       # in place: u becomes s
        else:
            h = geglu_fwd(u)[0]
    return u, h, _linear_bwd(x if x_saved is None else x_saved, weight, bias, True)


def _linear_bwd(x_in, weight: Tensor, bias: Optional[Tensor], need_dx: bool):
    def bwd(dy: Tensor, dx_add: Optional[Tensor] = None, geglu_u: Optional[Tensor] = None, geglu_s: Optional[Tensor] = None):
        """geglu_u = the [a | g] matrix whose GEGLU produced x (FeedForward): the returned gradient is then d/du [M, 2K], the GEGLU
        backward applied in the input-gradient GEMM's epilogue (nk_linear_dgrad_geglu).  geglu_s = the saved-derivative form
        [gelu(g) | a gelu'(g)] of linear_geglu_fwd(save_derivative=True) instead (nk_linear_dgrad_geglu_s)."""
        entry = "nk_linear_dgrad_geglu"
        if geglu_s is not None:
            if geglu_u is not None:
                raise ValueError("linear bwd: geglu_u and geglu_s are exclusive")
            geglu_u, entry = geglu_s, "nk_linear_dgrad_geglu_s"
        x = x_in() if callable(x_in) else x_in         # (selective recompute: the layer's input is rebuilt now, on the current stream)
        queued = _wgrad_queue is not None and _wgrad_queue.takes(weight) and dy.is_contiguous() and x.is_contiguous()
        # the bias gradient (column sums of dy) comes out of the weight-gradient launch: every parameter gradient is OVERWRITTEN by its
        # (single) producer unless accumulating -- the same mode for both
        db = grad_flat(bias) if bias is not None else None
        if queued:
            _wgrad_queue.add(dy, x, g2d(weight), db)
        else:
            on_wgrad_stream(lambda: gemm_tn_f32(dy, x, g2d(weight), wgrad_mode(weight), dbias=db), dy, x, owner=weight)
        if not need_dx:
            return None
        if geglu_u is not None:
            if dx_add is not None:
                raise ValueError("linear bwd: geglu_u and dx_add are exclusive")
            M, N = dy.shape
            K = weight.shape[1]
            _check2d(geglu_u, "geglu_u")
            if geglu_u.shape != (M, 2 * K):
                raise ValueError(f"linear bwd: geglu_u must be [{M}, {2 * K}], got {tuple(geglu_u.shape)}")
            du = torch.empty(M, 2 * K, dtype=BF16, device=dy.device)
            wq = w2d(weight)
            call(entry, dy.data_ptr(), wq.data_ptr(), geglu_u.data_ptr(), du.data_ptr(), M, N, K, dy.stride(0), wq.stride(0),
                 geglu_u.stride(0), du.stride(0), _stream())
            return du
        return gemm_nn(dy, w2d(weight), dx_add)

    return bwd


# ------------------------------------------------------------------------------------------------
# Conv2d (implicit GEMM)
# ------------------------------------------------------------------------------------------------
def _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, upsample) -> NkConvDesc:
    return NkConvDesc(N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, int(upsample))


def groupnorm_sums(x: Img, groups: int) -> Tensor:
    """[N, 2*groups] fp32 sums / sums of squares of x per (image, group): x.sums when its producer emitted them, else one pass over x."""
    if x.sums is not None and x.sums.shape == (x.N, 2 * groups):
        return x.sums
    sums = torch.empty(x.N, 2 * groups, dtype=torch.float32, device=x.t.device)
    ws = _ws(query("nk_groupnorm_ws_floats", x.N, x.H * x.W, x.C, groups), x.t.device)
    call("nk_groupnorm_sums", x.t.data_ptr(), sums.data_ptr(), ws.data_ptr(), x.N, x.H * x.W, x.C, groups, _stream())
    return sums


def conv2d_fwd(x: Img, weight: Tensor, bias: Optional[Tensor], stride: int = 1, padding=1, upsample: bool = False,
               rowvec: Optional[Tensor] = None, residual: Optional[Tensor] = None, need_dx: bool = True,
               asym_pad: bool = False, stats_groups: Optional[int] = None, cin_real: Optional[int] = None):
    """nn.Conv2d forward on channels-last data as implicit GEMM.
    padding: int (symmetric) ; asym_pad=True reproduces ConstantPad2d((0,1,0,1)) + padding 0 (model.py:71-79).
    rowvec: bf16 [N, Cout] added to every pixel of image n (ResBlock emb_out); residual: bf16 [N*Ho*Wo, Cout].
    stats_groups = G: the output Img carries `.sums` (its GroupNorm sums over G groups) when the kernel can emit them.
    cin_real = 3 | 4: x's 8 channels are 3 or 4 real ones plus zero padding (image / latent inputs): the forward of a plain 3 x 3
    stride-1 convolution then runs as the register-resident FMA kernel instead of a K = 72 implicit GEMM.
    bwd(dy) -> (dx Img | None, d_rowvec | None)."""
    Cout, Cin, KH, KW = weight.shape
    if x.C != Cin:
        raise ValueError(f"conv2d: input has {x.C} channels, weight expects {Cin}")
    Hin, Win = (2 * x.H, 2 * x.W) if upsample else (x.H, x.W)
    if asym_pad:
        pad_t = pad_l = 0
        Ho = (Hin + 1 - KH) // stride + 1
        Wo = (Win + 1 - KW) // stride + 1
    else:
        pad_t = pad_l = int(padding)
        Ho = (Hin + 2 * pad_t - KH) // stride + 1
        Wo = (Win + 2 * pad_l - KW) // stride + 1
    d = _conv_desc(x.N, x.H, x.W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo, upsample)
    y = torch.empty(x.N * Ho * Wo, Cout, dtype=BF16, device=x.t.device)
    _check2d(x.t, "x")
    if not x.t.is_contiguous():
        raise ValueError("conv2d: x must be dense channels-last")
    tiles = query("nk_conv2d_stats_tiles", C.byref(d), stats_groups) if stats_groups else 0
    sums_out = None
    if tiles:
        dev = x.t.device
        part = torch.empty(x.N, tiles, 2 * stats_groups, dtype=torch.float32, device=dev)
        call("nk_conv2d_fwd_stats", C.byref(d), x.t.data_ptr(), w2d(weight).data_ptr(), _p(bias), _p(rowvec), _p(residual), y.data_ptr(),
             part.data_ptr(), stats_groups, _stream())
        sums_out = torch.empty(x.N, 2 * stats_groups, dtype=torch.float32, device=dev)
        ws = _ws(query("nk_groupnorm_sums_ws_floats", x.N, tiles, stats_groups), dev)
        call("nk_groupnorm_sums_from_parts", part.data_ptr(), sums_out.data_ptr(), ws.data_ptr(), x.N, tiles, stats_groups, _stream())
    elif (cin_real in (3, 4) and Cin == 8 and KH == 3 and KW == 3 and stride == 1 and pad_t == 1 and not asym_pad and not upsample and rowvec is None
          and residual is None and Cout % 8 == 0):
        call("nk_conv3x3_few_channels_fwd", x.t.data_ptr(), w2d(weight).data_ptr(), _p(bias), y.data_ptr(), x.N, x.H, x.W, Cout, cin_real, _stream())
    else:
        call("nk_conv2d_fwd", C.byref(d), x.t.data_ptr(), w2d(weight).data_ptr(), _p(bias), _p(rowvec), _p(residual), y.data_ptr(), _stream())
    out = Img(y, x.N, Ho, Wo, sums_out)

    def bwd(dy: Tensor):
        _check2d(dy, "dy")
        if not dy.is_contiguous():
            raise ValueError("conv2d bwd: dy must be dense")
        acc = state_of(weight).grad_accumulate

        if weight.requires_grad:          # frozen convolutions (the LPIPS trunk) only pass the gradient through
            if bias is not None:       # the bias gradient rides in the weight-gradient launch
                on_wgrad_stream(lambda: call("nk_conv2d_wgrad_bias", C.byref(d), dy.data_ptr(), x.t.data_ptr(), g2d(weight).data_ptr(), grad_flat(bias).data_ptr(),
                                             wgrad_mode(weight), _stream()), dy, x.t, owner=weight)
            else:
                on_wgrad_stream(lambda: call("nk_conv2d_wgrad", C.byref(d), dy.data_ptr(), x.t.data_ptr(), g2d(weight).data_ptr(), wgrad_mode(weight), _stream()),
                                dy, x.t, owner=weight)
        drow = None
        if rowvec is not None:
            drow32 = torch.empty(x.N, Cout, dtype=torch.float32, device=dy.device)
            ws = _ws(x.N * query("nk_colsum_ws_floats", Ho * Wo, Cout), dy.device)
            call("nk_colsum_batched", dy.data_ptr(), drow32.data_ptr(), ws.data_ptr(), Ho * Wo, Cout, dy.stride(0), x.N, 0, _stream())   # one pair of launches, not one per image
            drow = cast_bf16(drow32)
        dx = None
        if need_dx:
            dxt = torch.empty(x.N * Hin * Win, Cin, dtype=BF16, device=dy.device)
            # the input gradient of a stride-1 3 x 3 "same" convolution is such a convolution of dy with the mirrored, channel-swapped
            # weights: where the halo-tile forward kernel takes that shape it runs there (one small transposing pass over the weights,
            # re-done per step because they change), otherwise on the transposed-operand gather kernel
            if query("nk_conv2d_dgrad_flipped_ok", C.byref(d)):
                wt = torch.empty(Cin * 9 * Cout, dtype=BF16, device=dy.device)
                call("nk_conv_weight_flip", w2d(weight).data_ptr(), wt.data_ptr(), Cout, Cin, 9, _stream())
                call("nk_conv2d_dgrad_flipped", C.byref(d), dy.data_ptr(), wt.data_ptr(), dxt.data_ptr(), _stream())
            else:
                call("nk_conv2d_dgrad", C.byref(d), dy.data_ptr(), w2d(weight).data_ptr(), dxt.data_ptr(), _stream())
            if upsample:
                dsm = torch.empty(x.N * x.H * x.W, Cin, dtype=BF16, device=dy.device)
                call("nk_upsample2x_bwd", dxt.data_ptr(), dsm.data_ptr(), x.N, x.H, x.W, Cin, _stream())
                dxt = dsm
            dx = Img(dxt, x.N, x.H, x.W)
        return dx, drow

    return out, bwd


# ------------------------------------------------------------------------------------------------
# norms / activations
# ------------------------------------------------------------------------------------------------
def groupnorm_fwd(x: Img, weight: Tensor, bias: Tensor, groups: int, eps: float, silu: bool):
    """GroupNorm (+SiLU).  bwd(dy, dx_add=None) -> dx token matrix."""
    if not x.t.is_contiguous():
        raise ValueError("groupnorm: x must be dense channels-last")
    N, HW, Cc = x.N, x.H * x.W, x.C
    y = torch.empty_like(x.t)
    mean = torch.empty(N, groups, dtype=torch.float32, device=x.t.device)
    rstd = torch.empty_like(mean)
    nws = query("nk_groupnorm_ws_floats", N, HW, Cc, groups)
    if x.sums is not None and x.sums.shape == (N, 2 * groups):
        # the producer's statistics epilogue already summed x: the normalisation pass alone
        call("nk_groupnorm_apply", x.t.data_ptr(), x.sums.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
             rstd.data_ptr(), N, HW, Cc, groups, float(eps), int(silu), _stream())
    else:
        ws = _ws(nws, x.t.device)
        call("nk_groupnorm_fwd", x.t.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
             ws.data_ptr(), N, HW, Cc, groups, float(eps), int(silu), _stream())

    def bwd(dy: Tensor, dx_add: Optional[Tensor] = None):
        dx = torch.empty_like(x.t)
        ws2 = _ws(nws, dy.device)
        call("nk_groupnorm_bwd", dy.data_ptr(), x.t.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
             _p(dx_add), dx.data_ptr(), grad_flat(weight).data_ptr(), grad_flat(bias).data_ptr(), ws2.data_ptr(), N, HW, Cc, groups,
             int(silu), int(state_of(weight).grad_accumulate), _stream())
        return dx

    return Img(y, x.N, x.H, x.W), bwd


def layernorm_fwd(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5):
    """LayerNorm over the last dim of a dense token matrix.  bwd(dy, dx_add=None) -> dx."""
    _check2d(x, "x")
    if not x.is_contiguous():
        raise ValueError("layernorm: x must be dense")
    M, Cc = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    call("nk_layernorm_fwd", x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, Cc, float(eps), _stream())

    def bwd(dy: Tensor, dx_add: Optional[Tensor] = None):
        dx = torch.empty_like(x)
        acc = state_of(weight).grad_accumulate
        if os.environ.get("NK_LN_FUSED", "1") != "0":
            # one pass over dy and x (round 5): dx and one row of dgamma / dbeta partials per workgroup; the rows are reduced later, off the
            # critical path -- with the block's other LayerNorms in ONE launch behind its batched weight gradients when a WgradQueue is
            # open (BasicTransformerBlock.bwd), else right away on the weight-gradient stream
            rows = query("nk_layernorm_part_rows", M)
            part = torch.empty(rows * 2 * Cc, dtype=torch.float32, device=dy.device)
            call("nk_layernorm_bwd_rows", dy.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dx_add),
                 dx.data_ptr(), part.data_ptr(), M, Cc, _stream())
            entry = (part, grad_flat(weight), grad_flat(bias), rows, Cc, bool(acc))
            if _wgrad_queue is not None:
                _wgrad_queue.add_colpart(*entry)
            else:
                on_wgrad_stream(lambda: colpart_reduce([entry]), part, owner=weight)
            return dx
        # NK_LN_FUSED=0: the three-kernel form of rounds 2-4 (dx on the main chain; a second pass over dy, x for the parameter gradients and
        # their column reduce on the weight-gradient stream)
        ws = _ws(query("nk_layernorm_ws_floats", M, Cc), dy.device)
        call("nk_layernorm_bwd_dx", dy.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dx_add),
             dx.data_ptr(), M, Cc, _stream())

        def params():   # gamma / beta gradients
            call("nk_layernorm_bwd_params", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                 grad_flat(weight).data_ptr(), grad_flat(bias).data_ptr(), ws.data_ptr(), M, Cc, int(acc), _stream())

        st_ = state_of(weight)
        if st_.norm_params_on_side_stream if st_.norm_params_on_side_stream is not None else st_.deferred is not None:
            on_wgrad_stream(params, dy, x, mean, rstd, ws, owner=weight)
        else:
            params()
        return dx

    return y, bwd


def geglu_fwd(u: Tensor):
    _check2d(u, "u")
    M, I2 = u.shape
    inner = I2 // 2
    y = torch.empty(M, inner, dtype=BF16, device=u.device)
    call("nk_geglu_fwd", u.data_ptr(), y.data_ptr(), M, inner, _stream())

    def bwd(dy: Tensor):
        du = torch.empty_like(u)
        call("nk_geglu_bwd", dy.data_ptr(), u.data_ptr(), du.data_ptr(), M, inner, _stream())
        return du

    return y, bwd


def gelu(x: Tensor, quick: bool = False) -> Tensor:
    """exact GELU, or x * sigmoid(1.702 x) with quick=True; forward only (frozen text encoders)"""
    y = torch.empty_like(x)
    call("nk_gelu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), int(quick), _stream())
    return y


def leaky_relu_fwd(x: Tensor, slope: float = 0.2):
    """(y, bwd); bwd(dy) -> dx"""
    y = torch.empty_like(x)
    call("nk_leaky_relu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), float(slope), _stream())

    def bwd(dy: Tensor) -> Tensor:
        dx = torch.empty_like(dy)
        call("nk_leaky_relu_bwd", dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), float(slope), _stream())
        return dx

    return y, bwd


def maxpool2x2_fwd(x: Img):
    """2x2 / stride 2 max pooling on channels-last tokens.  (y Img, bwd); bwd(dy tokens) -> dx tokens"""
    y = torch.empty(x.N * (x.H // 2) * (x.W // 2), x.C, dtype=BF16, device=x.t.device)
    call("nk_maxpool2x2_fwd", x.t.data_ptr(), y.data_ptr(), x.N, x.H, x.W, x.C, _stream())

    def bwd(dy: Tensor) -> Tensor:
        dx = torch.empty_like(x.t)
        call("nk_maxpool2x2_bwd", dy.data_ptr(), x.t.data_ptr(), dx.data_ptr(), x.N, x.H, x.W, x.C, _stream())
        return dx

    return Img(y, x.N, x.H // 2, x.W // 2), bwd


def maxpool_fwd(x: Img, kernel_size: int, stride: int):
    """nn.MaxPool2d(kernel_size, stride) (no padding, floor mode) on channels-last tokens.  (y Img, bwd); bwd(dy tokens) -> dx tokens"""
    if kernel_size == 2 and stride == 2 and x.H % 2 == 0 and x.W % 2 == 0:
        return maxpool2x2_fwd(x)
    Ho, Wo = (x.H - kernel_size) // stride + 1, (x.W - kernel_size) // stride + 1
    y = torch.empty(x.N * Ho * Wo, x.C, dtype=BF16, device=x.t.device)
    call("nk_maxpool_fwd", x.t.data_ptr(), y.data_ptr(), x.N, x.H, x.W, x.C, kernel_size, stride, _stream())

    def bwd(dy: Tensor) -> Tensor:
        dx = torch.empty_like(x.t)
        call("nk_maxpool_bwd", dy.data_ptr(), x.t.data_ptr(), dx.data_ptr(), x.N, x.H, x.W, x.C, kernel_size, stride, _stream())
        return dx

    return Img(y, x.N, Ho, Wo), bwd


def lpips_layer(f0: Img, f1: Img, w: Tensor, out: Tensor, accumulate: bool, eps: float = 1e-10):
    """out[n] (+)= LPIPS distance of one feature layer (see nk_lpips_layer_fwd); returns bwd(upstream[N]) -> d/d f1 tokens"""
    N, HW, Cc = f0.N, f0.H * f0.W, f0.C
    ws = _ws(query("nk_lpips_layer_ws_floats", N, HW), f0.t.device)
    call("nk_lpips_layer_fwd", f0.t.data_ptr(), f1.t.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), N, HW, Cc, float(eps), int(accumulate), _stream())

    def bwd(upstream: Tensor) -> Tensor:
        d = torch.empty_like(f1.t)
        call("nk_lpips_layer_bwd", f0.t.data_ptr(), f1.t.data_ptr(), w.data_ptr(), upstream.data_ptr(), d.data_ptr(), N, HW, Cc, float(eps), _stream())
        return d

    return bwd


def batchnorm_fwd(x: Tensor, weight: Tensor, bias: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor], eps: float = 1e-5,
                  momentum: float = 0.1, slope: float = 1.0):
    """nn.BatchNorm2d in training mode on a token matrix [N*H*W, C] (batch statistics per channel), with the LeakyReLU that follows
    it in the PatchGAN fused in (slope 1.0 = none).  Updates the running statistics in place.  bwd(dy) -> dx."""
    _check2d(x, "x")
    if not x.is_contiguous():
        raise ValueError("batchnorm: x must be dense")
    M, Cc = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    nws = query("nk_batchnorm_ws_floats", M, Cc)
    call("nk_batchnorm_fwd", x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(running_mean),
         _p(running_var), _ws(nws, x.device).data_ptr(), M, Cc, float(eps), float(momentum), float(slope), _stream())

    def bwd(dy: Tensor) -> Tensor:
        dx = torch.empty_like(x)
        call("nk_batchnorm_bwd", dy.data_ptr(), x.data_ptr(), y.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
             grad_flat(weight).data_ptr(), grad_flat(bias).data_ptr(), _ws(nws, x.device).data_ptr(), M, Cc, float(slope), int(state_of(weight).grad_accumulate),
             _stream())
        return dx

    return y, bwd


def batchnorm_eval(x: Tensor, weight: Tensor, bias: Tensor, running_mean: Tensor, running_var: Tensor, eps: float = 1e-5, slope: float = 1.0) -> Tensor:
    """nn.BatchNorm2d in evaluation mode on a token matrix [M, C] (running statistics), with the following LeakyReLU fused in."""
    _check2d(x, "x")
    if not x.is_contiguous():
        raise ValueError("batchnorm: x must be dense")
    M, Cc = x.shape
    y = torch.empty_like(x)
    call("nk_batchnorm_eval", x.data_ptr(), weight.data_ptr(), bias.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), y.data_ptr(),
         _ws(Cc + 64, x.device).data_ptr(), M, Cc, float(eps), float(slope), _stream())
    return y


def silu_fwd(x: Tensor):
    y = torch.empty_like(x)
    call("nk_silu_fwd", x.data_ptr(), y.data_ptr(), x.numel(), _stream())

    def bwd(dy: Tensor):
        dx = torch.empty_like(x)
        call("nk_silu_bwd", dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream())
        return dx

    return y, bwd


def cat_fwd(a: Img, b: Img):
    """torch.cat([a, b], dim=1) on channels-last images.  bwd(dout) -> (da, db) token matrices."""
    if (a.N, a.H, a.W) != (b.N, b.H, b.W):
        raise ValueError("cat: spatial shapes differ")
    rows = a.t.shape[0]
    out = torch.empty(rows, a.C + b.C, dtype=BF16, device=a.t.device)
    call("nk_cat_channels", a.t.data_ptr(), b.t.data_ptr(), out.data_ptr(), rows, a.C, b.C, _stream())
    Ca, Cb = a.C, b.C

    def bwd(dout: Tensor):
        da = torch.empty(rows, Ca, dtype=BF16, device=dout.device)
        db = torch.empty(rows, Cb, dtype=BF16, device=dout.device)
        call("nk_split_channels", dout.data_ptr(), da.data_ptr(), db.data_ptr(), rows, Ca, Cb, _stream())
        return da, db

    return Img(out, a.N, a.H, a.W), bwd


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def attention_fwd(q: Tensor, k: Tensor, v: Tensor, B: int, heads: int, dim_head: int, causal: bool = False, need_lse: bool = True,
                  return_lse: bool = False):
    """softmax(q k^T / sqrt(d)) v.  q [B*Lq, H*D], k/v [B*Lk, H*D] token matrices (column slices allowed).
    bwd(do) -> (dq, dk, dv) dense token matrices.  causal=True (frozen text transformers) is forward only.
    dim_head = 512 (the VAE mid block): csrc/attn512.h forward (need_lse=False skips the log-sum-exp output: inference), csrc/attn512_bwd.h backward."""
    for n, t in (("q", q), ("k", k), ("v", v)):
        _check2d(t, n)
    Lq, Lk = q.shape[0] // B, k.shape[0] // B
    HD = heads * dim_head
    o = torch.empty(B * Lq, HD, dtype=BF16, device=q.device)
    lse = torch.empty(B, heads, Lq, dtype=torch.float32, device=q.device) if (need_lse or dim_head != 512) else None
    d = NkAttnDesc()
    d.B, d.H, d.Lq, d.Lk, d.D = B, heads, Lq, Lk, dim_head
    d.sq, d.sk, d.sv, d.so = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    d.bq, d.bk, d.bv, d.bo = Lq * q.stride(0), Lk * k.stride(0), Lk * v.stride(0), Lq * o.stride(0)
    d.scale = float(dim_head) ** -0.5
    d.causal = int(causal)
    call("nk_attention_fwd", C.byref(d), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr() if lse is not None else None,
         _stream())

    def bwd(do: Tensor, dq: Optional[Tensor] = None, dk: Optional[Tensor] = None, dv: Optional[Tensor] = None):
        _check2d(do, "do")
        if dim_head > 160 and dim_head != 512:
            raise NotImplementedError("attention backward: head dim <= 160, or 512 (csrc/attn512_bwd.h)")
        if lse is None:
            raise ValueError("attention backward needs the forward's log-sum-exp: call attention_fwd(..., need_lse=True)")
        dq = torch.empty(B * Lq, HD, dtype=BF16, device=do.device) if dq is None else dq
        dk = torch.empty(B * Lk, HD, dtype=BF16, device=do.device) if dk is None else dk
        dv = torch.empty(B * Lk, HD, dtype=BF16, device=do.device) if dv is None else dv
        d.sdq, d.sdk, d.sdv, d.sdo = dq.stride(0), dk.stride(0), dv.stride(0), do.stride(0)
        d.bdq, d.bdk, d.bdv, d.bdo = Lq * dq.stride(0), Lk * dk.stride(0), Lk * dv.stride(0), Lq * do.stride(0)
        delta = _ws(query("nk_attention_bwd_ws_floats", C.byref(d)), do.device)
        call("nk_attention_bwd", C.byref(d), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(),
             dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), _stream())
        return dq, dk, dv

    if return_lse:              # [B, heads, Lq] fp32, natural log of the row sums of exp(scale q k^T) (the chunked recompute backward rebuilds P from it)
        return o, bwd, lse
    return o, bwd


def attention_unfused(q: Tensor, k: Tensor, v: Tensor, B: int) -> Tensor:
    """Single-head attention with any head dim via two MFMA GEMMs and a row softmax (inference only;
    the VAE mid block, modules/diffusion/model.py:224-243).  q/k/v dense [B*L, D]."""
    L = q.shape[0] // B
    D = q.shape[1]
    out = torch.empty_like(q)
    s = torch.empty(L, L, dtype=BF16, device=q.device)
    for b in range(B):
        sl = slice(b * L, (b + 1) * L)
        gemm_nt(q[sl], k[sl], alpha=float(D) ** -0.5, out=s)
        call("nk_softmax_rows", s.data_ptr(), L, L, _stream())
        gemm_nn(s, v[sl], out=out[sl])
    return out


ATTN512_FLASH_MAX_L = 2048    # tokens per sample up to which the flash backward kernels are the faster form (NK_ATTN512_BWD=auto)
ATTN512_BWD_CHUNK = 2048      # query rows per backward chunk: the recomputed score / probability block is [chunk, L] bf16 (64 MB at L = 16384)


def attention512_fwd(q: Tensor, k: Tensor, v: Tensor, B: int):
    """Single-head attention of head dim 512 WITH a backward (the VAE mid block when the autoencoder is trained: modules/diffusion/model.py:224-243,
    models/autoencoder.py:280-293).  Forward: the one-kernel flash forward of csrc/attn512.h -- nothing of size L x L is kept (rounds 2-4 kept
    the [B, L, L] probabilities of a two-GEMM forward).  Backward, two forms (round 5):
      flash     -- csrc/attn512_bwd.h: one kernel template for dQ (32 queries per workgroup) and dK / dV (32 keys per workgroup), the scores
                   recomputed tile by tile in registers from the forward's log-sum-exp; nothing of size L x L anywhere;
      recompute -- per sample and per chunk of ATTN512_BWD_CHUNK query rows the probabilities are rebuilt through HBM (scores GEMM + row
                   softmax) and consumed at once by the four gradient GEMMs (dP = dO V^T, dV += P^T dO, dS = P o (dP - rowsum(P o dP)) / sqrt(D),
                   dQ = dS K, dK += dS^T Q): working set [chunk, L].
    q / k / v dense [B*L, D]; bwd(do) -> (dq, dk, dv)."""
    L, D = q.shape[0] // B, q.shape[1]
    if D != 512:
        raise ValueError(f"attention512_fwd: head dim 512 only, got {D} (head dims <= 160: attention_fwd)")
    scale = float(D) ** -0.5
    # NK_ATTN512_BWD: "auto" (default) = the flash backward kernels (csrc/attn512_bwd.h: scores recomputed tile by tile in registers from the
    # forward's log-sum-exp, nothing in HBM) up to ATTN512_FLASH_MAX_L tokens, the chunked recompute beyond; "1" / "0" force one or the other.
    # Measured (tools/bench_attn512_bwd.py, batch 4): L = 1024 (config 5) 250 vs 449 us; L = 4096 1.71 vs 1.17 ms; L = 16384 (batch 1) 6.6 vs
    # 2.8 ms -- the flash kernels are correct and lean, not tuned (one wave per SIMD, register-staged tiles): the tile engine's GEMMs win once
    # the [chunk, L] scratch is large enough to run them at speed.
    mode = os.environ.get("NK_ATTN512_BWD", "auto")
    if mode == "1" or (mode == "auto" and L <= ATTN512_FLASH_MAX_L):
        out, b_att = attention_fwd(q, k, v, B, 1, D, need_lse=True)
        return out, (lambda do: b_att(do))
    # (nk_softmax_rows takes rows of whole 16-byte chunks only -- say so HERE, not in a backward that follows a forward which
    # accepted the shape.  The recompute form rebuilds P from bf16-ROUNDED scores (gemm_nt writes bf16), not from the flash forward's fp32
    # scores: at logits of 30-50 the two differ by several per cent per element -- INTEGRATION.md section 6, NK_ATTN512_BWD.)
    if L % 8:
        raise ValueError(f"attention512_fwd: the chunked recompute backward (L = {L} > {ATTN512_FLASH_MAX_L} tokens per sample, or NK_ATTN512_BWD=0) "
                         f"needs L % 8 == 0; NK_ATTN512_BWD=1 selects the flash backward, which takes any length")
    out, _, lse = attention_fwd(q, k, v, B, 1, D, need_lse=True, return_lse=True)
    return out, _attention_recompute_bwd(q, k, v, B, lse=lse.reshape(B * L))


def attention_anydim_fwd(q: Tensor, k: Tensor, v: Tensor, B: int):
    """Single-head attention of ANY head dim (a multiple of 8) WITH a backward: the reference's AttnBlock trains at whatever channel count the
    autoencoder's last level has (modules/diffusion/model.py:224-243); head dims <= 160 and 512 have flash kernels (attention_fwd, attention512_fwd),
    everything else -- a VAE with ch * ch_mult[-1] of 256 or 384 -- takes this form: the two-GEMM forward of attention_unfused and
    the chunked recompute backward shared with attention512_fwd.  Working set [L, L] bf16 forward, [chunk, L] backward.  q / k / v dense [B*L, D]."""
    L, D = q.shape[0] // B, q.shape[1]
    if D % 8 or L % 8:
        raise ValueError(f"attention_anydim_fwd: head dim and tokens per sample must be multiples of 8, got D = {D}, L = {L}")
    return attention_unfused(q, k, v, B), _attention_recompute_bwd(q, k, v, B)


def _attention_recompute_bwd(q: Tensor, k: Tensor, v: Tensor, B: int, lse: Optional[Tensor] = None):
    """bwd(do) -> (dq, dk, dv) of single-head attention by recomputing the probabilities chunk by chunk (attention512_fwd has the formulae).

    lse ([B*L] fp32: the flash forward's log-sum-exp per query row) makes the rebuilt probabilities follow the forward's.  The scores
    GEMM writes bf16; a score of 40 has a bf16 spacing of 0.25, so exp() of the ROUNDED score is off by up to 13 % -- while the forward exponentiated
    fp32 scores.  With lse the row's log-sum-exp is subtracted INSIDE the GEMM, before the rounding: q gets eight more columns holding -lse / scale as
    three bf16 pieces (exact to 2^-24 relative) and k eight more of ones, so the accumulator holds scale q.k - lse in fp32 and what is rounded to bf16
    is log p: <= 0, with |log p| <= 4 for every entry that carries weight -- spacing <= 2^-6, 1.6 % at worst, typically 0.4 %.  nk_softmax_rows then
    renormalises the row (its sum is 1 up to those roundings)."""
    L, D = q.shape[0] // B, q.shape[1]
    scale = float(D) ** -0.5
    if lse is not None:
        neg = -lse.to(torch.float32) / scale
        p1 = neg.to(BF16)
        r1 = neg - p1.float()
        p2 = r1.to(BF16)
        p3 = (r1 - p2.float()).to(BF16)
        q_aug = torch.cat([q, torch.stack([p1, p2, p3], dim=1), torch.zeros(q.shape[0], 5, dtype=BF16, device=q.device)], dim=1)
        ones = torch.zeros(k.shape[0], 8, dtype=BF16, device=k.device)
        ones[:, :3] = 1.0
        k_aug = torch.cat([k, ones], dim=1)
    else:
        q_aug, k_aug = q, k

    def bwd(do: Tensor):
        _check2d(do, "do")
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dk32 = torch.empty(L, D, dtype=torch.float32, device=q.device)
        dv32 = torch.empty(L, D, dtype=torch.float32, device=q.device)
        for b in range(B):
            for c0 in range(0, L, ATTN512_BWD_CHUNK):
                rows = slice(b * L + c0, b * L + min(c0 + ATTN512_BWD_CHUNK, L))
                keys = slice(b * L, (b + 1) * L)
                n = rows.stop - rows.start
                p = gemm_nt(q_aug[rows], k_aug[keys], alpha=scale)                 # scores [n, L] (minus the row's log-sum-exp where known) ...
                call("nk_softmax_rows", p.data_ptr(), n, L, _stream())             # ... -> probabilities, as the forward's
                dp = gemm_nt(do[rows], v[keys])                                     # dP = dO V^T
                gemm_tn_f32(p, do[rows], dv32, c0 > 0)                              # dV (+)= P^T dO
                call("nk_softmax_rows_bwd", p.data_ptr(), dp.data_ptr(), n, L, scale, _stream())    # dP -> dS (scaled), in place
                gemm_nn(dp, k[keys], out=dq[rows])                                  # dQ = dS K
                gemm_tn_f32(dp, q[rows], dk32, c0 > 0)                              # dK (+)= dS^T Q
            dv[b * L:(b + 1) * L].copy_(cast_bf16(dv32))
            dk[b * L:(b + 1) * L].copy_(cast_bf16(dk32))
        return dq, dk, dv

    return bwd


# ------------------------------------------------------------------------------------------------
# misc
# ------------------------------------------------------------------------------------------------
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    tf = t.to(torch.float32).contiguous()
    out = torch.empty(tf.shape[0], dim, dtype=BF16, device=t.device)
    call("nk_timestep_embedding", tf.data_ptr(), out.data_ptr(), tf.shape[0], dim, float(max_period), _stream())
    return out
