"""Building blocks shared by the module mirrors: the autograd bridge, the Conv2d leaf, the flat
parameter store (fp32 masters / bf16 shadows / fp32 grads in three contiguous HBM buffers).
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Optional, Sequence

import torch
from torch import Tensor, nn

from . import ops
from .lib import call
from .ops import BF16, Img


# ------------------------------------------------------------------------------------------------
# autograd bridge: one node per top-level module call; everything inside is our own fwd/bwd chain
# ------------------------------------------------------------------------------------------------
class NkFunction(torch.autograd.Function):
    """forward(run, n_in, *tensor_inputs, *params): `run(*tensor_inputs)` returns (outputs, bwd).
    Parameters ride along only so the output requires grad; their gradients are written in place into
    param.grad by the HIP kernels, so backward returns None for them."""

    @staticmethod
    def forward(ctx, run: Callable, n_in: int, *args):
        with ops.recording_backward():      # (grad mode is off in here: the modules ask ops.recording() whether a backward follows)
            out, bwd = run(*args[:n_in])
        ctx.bwd = bwd
        ctx.n_in = n_in
        ctx.n_args = len(args)
        ctx.in_meta = [(a.dtype, a.shape) if isinstance(a, Tensor) else None for a in args[:n_in]]
        return out

    @staticmethod
    def backward(ctx, *gouts):
        bwd = ctx.bwd
        ctx.bwd = None  # free saved activations as soon as they are consumed
        if bwd is None:
            raise RuntimeError("neurosis_amd: backward called twice on the same graph (activations already freed)")
        gins = bwd(*gouts)
        ops.join_wgrad_stream()  # parameter gradients written on the side stream are visible to whatever runs next
        if not isinstance(gins, tuple):
            gins = (gins,)
        fixed = []
        for g, meta in zip(gins, ctx.in_meta):
            if g is not None and meta is not None and g.dtype != meta[0]:
                g = g.to(meta[0])
            fixed.append(g)
        fixed += [None] * (ctx.n_in - len(fixed))
        return (None, None, *fixed, *([None] * (ctx.n_args - ctx.n_in)))


def apply_module(run: Callable, inputs: Sequence[Optional[Tensor]], module: nn.Module):
    params = [p for p in module.parameters() if p.requires_grad]
    if not torch.is_grad_enabled() or not (params or any(isinstance(t, Tensor) and t.requires_grad for t in inputs)):
        out, _ = run(*inputs)
        return out
    return NkFunction.apply(run, len(inputs), *inputs, *params)


def as_tokens(x: Tensor) -> Tensor:
    """[.., C] tensor of any float dtype -> dense bf16 token matrix [prod(..), C] (C % 8 == 0)."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype == BF16 and x2.is_contiguous():
        return x2
    if x2.dtype == torch.float32:
        return ops.cast_bf16(x2.contiguous())
    if x2.dtype == BF16:
        return x2.contiguous()
    raise ValueError(f"unsupported dtype {x2.dtype}")


def grad_to_tokens(g: Tensor, C: int) -> Tensor:
    """incoming autograd gradient of a logical-NCHW output -> dense channels-last token matrix."""
    return Img.from_nchw(g).t if g.dim() == 4 else as_tokens(g)


# ------------------------------------------------------------------------------------------------
# leaves
# ------------------------------------------------------------------------------------------------
def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


class Conv2d(nn.Module):
    """nn.Conv2d replacement (same parameter names / OIHW shapes) running as implicit GEMM on channels-last
    bf16 data.  Channel counts that are not multiples of 8 (the 4-channel latent, the 3-channel image) run
    through zero-padded weight shadows; such inputs must arrive padded and such outputs leave padded."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0, bias: bool = True, asym_pad: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.asym_pad = kernel_size, stride, padding, asym_pad
        self.weight = ops.conv_weight_param(out_channels, in_channels, kernel_size, kernel_size)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._pad_cache = None

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size * self.kernel_size
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    @property
    def padded(self) -> bool:
        return self.in_channels % 8 != 0 or self.out_channels % 8 != 0

    def _padded_params(self):
        """(weight, bias) stand-ins with channels padded to multiples of 8 (tiny tensors, rebuilt per param epoch)."""
        stamp = (ops._param_stamp(self.weight), None if self.bias is None else ops._param_stamp(self.bias), self.weight.requires_grad)
        cache = self._pad_cache
        same = cache is not None and cache[1].device == self.weight.device and cache[0][2] == stamp[2]
        # Under a hipGraph capture the stand-ins are ALWAYS refilled, in place: a replay runs none of this Python, so the refill
        # has to be part of every captured forward (a graph captured while the cache happened to be valid would read last
        # step's weights for ever -- or freed memory), and the buffers have to stay where the other graphs expect them.
        if same and cache[0] == stamp and not ops.capturing():
            return cache[1], cache[2]
        if same:
            wp, bp = cache[1], cache[2]
            with torch.no_grad():
                wv = wp.data.permute(0, 2, 3, 1)
                wv.zero_()
                wv[: self.out_channels, :, :, : self.in_channels].copy_(self.weight.detach().permute(0, 2, 3, 1))
                if bp is not None:
                    bp.data.zero_()
                    bp.data[: self.out_channels].copy_(self.bias.detach())
            # (writes through .data do not move the tensors' version counters, which is what ops.shadow goes by: say so)
            wp._nk_shadow_stamp = None
            if bp is not None:
                bp._nk_shadow_stamp = None
            self._pad_cache = (stamp, wp, bp)
            return wp, bp
        ci, co, k = _pad8(self.in_channels), _pad8(self.out_channels), self.kernel_size
        with torch.no_grad():
            w = torch.zeros(co, k, k, ci, device=self.weight.device)
            w[: self.out_channels, :, :, : self.in_channels] = self.weight.detach().permute(0, 2, 3, 1)
            wp = nn.Parameter(w.permute(0, 3, 1, 2), requires_grad=self.weight.requires_grad)   # (frozen convs skip their weight gradient)
            bp = None
            if self.bias is not None:
                b = torch.zeros(co, device=self.weight.device)
                b[: self.out_channels] = self.bias.detach()
                bp = nn.Parameter(b, requires_grad=self.weight.requires_grad)
        # the stand-ins always overwrite their own small gradient (folded back below) but share the engine's side stream
        st = ops.state_of(self.weight).derived()
        wp._nk_state = st
        if bp is not None:
            bp._nk_state = st
        self._pad_cache = (stamp, wp, bp)
        return wp, bp

    def fwd(self, x: Img, rowvec: Optional[Tensor] = None, residual: Optional[Tensor] = None, upsample: bool = False, need_dx: bool = True,
            stats_groups: Optional[int] = None):
        """stats_groups = G: ask for the output's GroupNorm sums (out.sums) -- see ops.conv2d_fwd."""
        if not self.padded:
            return ops.conv2d_fwd(x, self.weight, self.bias, self.stride, self.padding, upsample, rowvec, residual, need_dx, self.asym_pad,
                                  stats_groups=stats_groups)
        wp, bp = self._padded_params()
        wp.grad = None
        if bp is not None:
            bp.grad = None
        out, b = ops.conv2d_fwd(x, wp, bp, self.stride, self.padding, upsample, rowvec, residual, need_dx, self.asym_pad,
                                cin_real=self.in_channels if self.in_channels in (3, 4) else None)

        def bwd(dy: Tensor):
            acc = ops.state_of(self.weight).grad_accumulate
            res = b(dy)
            if not self.weight.requires_grad:
                return res
            ops.join_wgrad_stream(wp)
            with torch.no_grad():  # fold the padded gradient back (a few thousand elements)
                g = wp.grad[: self.out_channels, : self.in_channels]
                ops.grad_flat(self.weight)
                ops.grad_flat(self.bias) if self.bias is not None else None
                if acc:
                    self.weight.grad.add_(g)
                    if self.bias is not None:
                        self.bias.grad.add_(bp.grad[: self.out_channels])
                else:   # like every other parameter gradient: overwritten by its producer unless accumulating
                    self.weight.grad.copy_(g)
                    if self.bias is not None:
                        self.bias.grad.copy_(bp.grad[: self.out_channels])
            return res

        return out, bwd

    def forward(self, x: Tensor) -> Tensor:
        def run(x):
            N, _, H, W = x.shape
            if self.in_channels % 8:
                img = Img(ops.nchw_to_tokens(x, _pad8(self.in_channels)), N, H, W)
            else:
                img = Img.from_nchw(x)
            out, bwd = self.fwd(img)

            def bwd2(g):
                gt = grad_to_tokens(g, out.C)
                if out.C != self.out_channels:
                    raise NotImplementedError("backward through a channel-padded conv output is only supported inside UNetModel")
                dx, _ = bwd(gt)
                if self.in_channels % 8:
                    return ops.tokens_to_nchw(dx.t, N, self.in_channels, H, W, dtype=x.dtype)
                return dx.to_nchw()

            y = out.to_nchw()
            if out.C != self.out_channels:
                y = ops.tokens_to_nchw(out.t, N, self.out_channels, out.H, out.W, dtype=BF16)
            return y, bwd2

        return apply_module(run, [x], self)


def linear_module_fwd(lin: nn.Linear, x: Tensor, residual: Optional[Tensor] = None, need_dx: bool = True, x_saved=None):
    return ops.linear_fwd(x, lin.weight, lin.bias, residual, need_dx, x_saved)


# ------------------------------------------------------------------------------------------------
# flat parameter store
# ------------------------------------------------------------------------------------------------
class FlatParamStore:
    """Re-homes a model's parameters into three flat HBM buffers laid out in registration order:
    fp32 masters, bf16 shadows (what the kernels read) and fp32 gradients (what the kernels write, what
    the data-parallel all-reduce moves, what the fused optimizer reads).  param.data / param.grad become
    views, so state_dict keys, shapes and checkpoint loading are unchanged."""

    ALIGN = 64  # elements: keeps every view 128 B (bf16) / 256 B (fp32) aligned

    def __init__(self, params: Iterable[nn.Parameter]):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatParamStore: no parameters")
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = total
        self.master = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(total, dtype=BF16, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg: Optional[Tensor] = None
        self.exp_avg_sq: Optional[Tensor] = None
        self.step_count = 0
        self.epoch = 0          # bumped whenever THIS store's masters / shadows change (ops.shadow compares against it)
        self.listeners = []     # objects with masters_changed(): state derived from the masters (Adafactor RMS sums, EMA seed)
        # backward-health word: kernels that detect an unusable result (a stream-K fix-up that gave up) raise it; the fused
        # optimizer kernels refuse to apply an update while it is set (models/diffusion.DiffusionEngine.optimizer_step)
        self.state = ops.EngineState()
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                p._nk_index = i
                self.master[off:off + n].copy_(ops._phys_flat(p))
                p.data = self._view(self.master, off, p)
                p.grad = self._view(self.grad, off, p)
                p._nk_shadow = self.shadow[off:off + n]
                p._nk_store = self
                p._nk_offset = off

        self.refresh()

    @staticmethod
    def _view(flat: Tensor, off: int, p: Tensor) -> Tensor:
        n = p.numel()
        if p.dim() == 4:
            O, I, KH, KW = p.shape
            return flat[off:off + n].view(O, KH, KW, I).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    def add_listener(self, obj) -> None:
        if all(o is not obj for o in self.listeners):
            self.listeners.append(obj)

    def refresh(self) -> None:
        """The fp32 masters were changed from OUTSIDE the fused optimizers (load_state_dict, broadcast, EMA swap): recompute
        every bf16 shadow and tell whoever keeps state derived from the masters."""
        call("nk_cast_f32_to_bf16", self.master.data_ptr(), self.shadow.data_ptr(), self.numel, ops._stream())
        self._mark_fresh()
        for o in self.listeners:
            o.masters_changed()

    masters_changed = refresh

    def _mark_fresh(self) -> None:
        """Shadows are current (a fused optimizer step rewrote them together with the masters, or refresh() did)."""
        self.epoch += 1
        ops.state.param_epoch += 1      # global "some parameter changed" counter: keys of captured graphs

    def zero_grad(self) -> None:
        """Not needed between steps: every parameter gradient is overwritten by its producer on the first micro-batch
        (store.state.grad_accumulate False) and added to on the following ones.  Kept for callers that skip parameters."""
        self.grad.zero_()

    def adamw_step(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, grad_scale: float = 1.0) -> None:
        """One fused AdamW update over the whole flat buffer; rewrites the bf16 shadows in the same pass."""
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.master)
            self.exp_avg_sq = torch.zeros_like(self.master)
        self.step_count += 1
        call("nk_adamw_flat", self.master.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
             self.shadow.data_ptr(), self.numel, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
             int(self.step_count), float(grad_scale), ops._stream())
        self._mark_fresh()

    # -- checkpointing of the fused AdamW state (torch.optim.AdamW's per-parameter keys) -----------------
    def optimizer_state_dict(self) -> dict:
        state = {}
        if self.exp_avg is not None:
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self._view(self.exp_avg, off, p).detach().clone().contiguous(),
                            "exp_avg_sq": self._view(self.exp_avg_sq, off, p).detach().clone().contiguous()}
        return {"state": state, "param_groups": [{"params": list(range(len(self.params)))}]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        state = sd.get("state", {})
        if not state:
            self.exp_avg = self.exp_avg_sq = None
            self.step_count = 0
            return
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.master)
            self.exp_avg_sq = torch.zeros_like(self.master)
        steps = set()
        with torch.no_grad():
            for i, st in state.items():
                p, off = self.params[int(i)], self.offsets[int(i)]
                self._view(self.exp_avg, off, p).copy_(st["exp_avg"])
                self._view(self.exp_avg_sq, off, p).copy_(st["exp_avg_sq"])
                steps.add(int(st["step"]))
        if len(steps) != 1:
            raise ValueError(f"FlatParamStore.load_optimizer_state_dict: per-parameter step counts differ ({sorted(steps)})")
        self.step_count = steps.pop()

    def param_range(self, module: nn.Module) -> tuple[int, int]:
        """[lo, hi) element range of the flat buffers covered by `module`'s parameters."""
        offs = [(p._nk_offset, p._nk_offset + p.numel()) for p in module.parameters() if getattr(p, "_nk_store", None) is self]
        if not offs:
            return (0, 0)
        return (min(o[0] for o in offs), max(o[1] for o in offs))


def adjacent(*params: Tensor) -> bool:
    """True if the bf16 shadows (and hence grads) of `params` are back to back in one flat store."""
    if any(getattr(p, "_nk_store", None) is None for p in params):
        return False
    if len({id(p._nk_store) for p in params}) != 1:
        return False
    for a, b in zip(params[:-1], params[1:]):
        if a._nk_offset + a.numel() != b._nk_offset:
            return False
    return True
