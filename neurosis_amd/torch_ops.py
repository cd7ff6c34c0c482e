"""`torch.library` registration of the HIP kernels: namespace `neurosis_hip` (SURVEY.md section 8(b) "C-ABI / op surface",
north_star "exposed as PyTorch-ROCm custom ops").

The training step itself does not go through the dispatcher -- the UNet is one explicit forward/backward chain over the
C-ABI -- but every kernel family is ALSO a dispatcher-visible op here, so a caller outside this package
(the reference's own modules, `torch.compile`, a C++ host of libtorch) can use them one by one:

    torch.ops.neurosis_hip.linear(x, w, bias)                      differentiable (autograd -> _dgrad / _wgrad ops)
    torch.ops.neurosis_hip.layernorm(x, gamma, beta, eps)           "
    torch.ops.neurosis_hip.groupnorm_silu(x, gamma, beta, N, groups, eps, silu)
    torch.ops.neurosis_hip.geglu(u)                                 "
    torch.ops.neurosis_hip.attention(q, k, v, B, heads)             "
    torch.ops.neurosis_hip.conv2d(x, w, bias, N, H, W, stride, padding)     channels-last tokens in / out
    ... and the raw `*_fwd / *_dgrad / *_wgrad / *_bwd` ops they are made of, `timestep_embedding`, `nchw_to_nlc`, `nlc_to_nchw`.

Dispatch keys: CUDA (= HIP on ROCm) only, plus Meta kernels (shapes / dtypes, so FakeTensor tracing and `torch.compile`
work).  There is deliberately NO CPU key: the product has no CPU path (the oracle under oracle/ is test infrastructure), so a
CPU tensor gets torch's own "not implemented for CPU" error.  Conventions are those of neurosis_amd.ops: bf16 token matrices
[rows, C] with unit inner stride, fp32 parameters cast by the caller (weights here are bf16 operands), fp32 statistics.

Reference call sites per op: include/neurosis_hip.h (each C entry point cites them) and INTEGRATION.md section 5.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .ops import BF16, Img

NS = "neurosis_hip"
_CUDA = "cuda"


def _op(name: str, mutates=()):
    return torch.library.custom_op(f"{NS}::{name}", mutates_args=mutates, device_types=_CUDA)


# ---------------------------------------------------------------------------------------------------------------
# Linear: y = alpha * x @ w^T + bias + residual
# ---------------------------------------------------------------------------------------------------------------
@_op("linear_fwd")
def linear_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None, alpha: float = 1.0) -> Tensor:
    return ops.gemm_nt(x, w, bias, residual, alpha)


@linear_fwd.register_fake
def _(x, w, bias=None, residual=None, alpha=1.0):
    return x.new_empty(x.shape[0], w.shape[0])


@_op("linear_dgrad")
def linear_dgrad(dy: Tensor, w: Tensor) -> Tensor:
    return ops.gemm_nn(dy, w)


@linear_dgrad.register_fake
def _(dy, w):
    return dy.new_empty(dy.shape[0], w.shape[1])


@_op("linear_wgrad")
def linear_wgrad(dy: Tensor, x: Tensor) -> Tensor:
    dw = torch.empty(dy.shape[1], x.shape[1], dtype=torch.float32, device=dy.device)
    ops.gemm_tn_f32(dy, x, dw, False)
    return dw


@linear_wgrad.register_fake
def _(dy, x):
    return dy.new_empty(dy.shape[1], x.shape[1], dtype=torch.float32)


@_op("colsum")
def colsum(dy: Tensor) -> Tensor:
    out = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device)
    ops.colsum(dy, out, False)
    return out


@colsum.register_fake
def _(dy):
    return dy.new_empty(dy.shape[1], dtype=torch.float32)


@_op("linear")
def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    return ops.gemm_nt(x, w, bias)


@linear.register_fake
def _(x, w, bias=None):
    return x.new_empty(x.shape[0], w.shape[0])


def _linear_setup(ctx, inputs, output):
    x, w, bias = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None


def _linear_bwd(ctx, dy):
    x, w = ctx.saved_tensors
    dy = dy.contiguous()
    dx = torch.ops.neurosis_hip.linear_dgrad(dy, w) if ctx.needs_input_grad[0] else None
    dw = torch.ops.neurosis_hip.linear_wgrad(dy, x).to(w.dtype) if ctx.needs_input_grad[1] else None
    db = torch.ops.neurosis_hip.colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
    return dx, dw, db


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ---------------------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------------------
@_op("layernorm_fwd")
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tuple[Tensor, Tensor, Tensor]:
    M, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(M, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ops.call("nk_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, C, float(eps), ops._stream())
    return y, mean, rstd


@layernorm_fwd.register_fake
def _(x, gamma, beta, eps=1e-5):
    return torch.empty_like(x), x.new_empty(x.shape[0], dtype=torch.float32), x.new_empty(x.shape[0], dtype=torch.float32)


@_op("layernorm_bwd")
def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    M, C = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    ws = ops._ws(ops.query("nk_layernorm_ws_floats", M, C), x.device)
    ops.call("nk_layernorm_bwd", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, dx.data_ptr(), dg.data_ptr(),
             db.data_ptr(), ws.data_ptr(), M, C, 0, ops._stream())
    return dx, dg, db


@layernorm_bwd.register_fake
def _(dy, x, gamma, mean, rstd):
    return torch.empty_like(x), gamma.new_empty(gamma.shape, dtype=torch.float32), gamma.new_empty(gamma.shape, dtype=torch.float32)


@_op("layernorm")
def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    return torch.ops.neurosis_hip.layernorm_fwd(x, gamma, beta, eps)[0]


@layernorm.register_fake
def _(x, gamma, beta, eps=1e-5):
    return torch.empty_like(x)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    _, mean, rstd = torch.ops.neurosis_hip.layernorm_fwd(x, gamma, beta, eps)     # statistics for backward (x is kept, not y)
    ctx.save_for_backward(x, gamma, mean, rstd)


def _ln_bwd(ctx, dy):
    x, gamma, mean, rstd = ctx.saved_tensors
    dx, dg, db = torch.ops.neurosis_hip.layernorm_bwd(dy.contiguous(), x, gamma, mean, rstd)
    return dx, dg, db, None


layernorm.register_autograd(_ln_bwd, setup_context=_ln_setup)


# ---------------------------------------------------------------------------------------------------------------
# GroupNorm (+SiLU) on channels-last tokens [N*HW, C]
# ---------------------------------------------------------------------------------------------------------------
@_op("groupnorm_silu_fwd")
def groupnorm_silu_fwd(x: Tensor, gamma: Tensor, beta: Tensor, N: int, groups: int, eps: float, silu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    HW, C = x.shape[0] // N, x.shape[1]
    y = torch.empty_like(x)
    mean = torch.empty(N, groups, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    ws = ops._ws(ops.query("nk_groupnorm_ws_floats", N, HW, C, groups), x.device)
    ops.call("nk_groupnorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), N, HW, C, groups,
             float(eps), int(silu), ops._stream())
    return y, mean, rstd


@groupnorm_silu_fwd.register_fake
def _(x, gamma, beta, N, groups, eps, silu):
    return torch.empty_like(x), x.new_empty(N, groups, dtype=torch.float32), x.new_empty(N, groups, dtype=torch.float32)


@_op("groupnorm_silu_bwd")
def groupnorm_silu_bwd(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, rstd: Tensor, N: int, groups: int, silu: bool) -> Tuple[Tensor, Tensor, Tensor]:
    HW, C = x.shape[0] // N, x.shape[1]
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    ws = ops._ws(ops.query("nk_groupnorm_ws_floats", N, HW, C, groups), x.device)
    ops.call("nk_groupnorm_bwd", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), None, dx.data_ptr(), dg.data_ptr(),
             db.data_ptr(), ws.data_ptr(), N, HW, C, groups, int(silu), 0, ops._stream())
    return dx, dg, db


@groupnorm_silu_bwd.register_fake
def _(dy, x, gamma, beta, mean, rstd, N, groups, silu):
    return torch.empty_like(x), gamma.new_empty(gamma.shape, dtype=torch.float32), gamma.new_empty(gamma.shape, dtype=torch.float32)


@_op("groupnorm_silu")
def groupnorm_silu(x: Tensor, gamma: Tensor, beta: Tensor, N: int, groups: int, eps: float, silu: bool) -> Tensor:
    return torch.ops.neurosis_hip.groupnorm_silu_fwd(x, gamma, beta, N, groups, eps, silu)[0]


@groupnorm_silu.register_fake
def _(x, gamma, beta, N, groups, eps, silu):
    return torch.empty_like(x)


def _gn_setup(ctx, inputs, output):
    x, gamma, beta, N, groups, eps, silu = inputs
    _, mean, rstd = torch.ops.neurosis_hip.groupnorm_silu_fwd(x, gamma, beta, N, groups, eps, silu)
    ctx.save_for_backward(x, gamma, beta, mean, rstd)
    ctx.meta = (N, groups, silu)


def _gn_bwd(ctx, dy):
    x, gamma, beta, mean, rstd = ctx.saved_tensors
    N, groups, silu = ctx.meta
    dx, dg, db = torch.ops.neurosis_hip.groupnorm_silu_bwd(dy.contiguous(), x, gamma, beta, mean, rstd, N, groups, silu)
    return dx, dg, db, None, None, None, None


groupnorm_silu.register_autograd(_gn_bwd, setup_context=_gn_setup)


# ---------------------------------------------------------------------------------------------------------------
# GEGLU: y = u[:, :I] * gelu_erf(u[:, I:])
# ---------------------------------------------------------------------------------------------------------------
@_op("geglu_fwd")
def geglu_fwd(u: Tensor) -> Tensor:
    return ops.geglu_fwd(u)[0]


@geglu_fwd.register_fake
def _(u):
    return u.new_empty(u.shape[0], u.shape[1] // 2)


@_op("geglu_bwd")
def geglu_bwd(dy: Tensor, u: Tensor) -> Tensor:
    du = torch.empty_like(u)
    ops.call("nk_geglu_bwd", dy.data_ptr(), u.data_ptr(), du.data_ptr(), u.shape[0], u.shape[1] // 2, ops._stream())
    return du


@geglu_bwd.register_fake
def _(dy, u):
    return torch.empty_like(u)


@_op("geglu")
def geglu(u: Tensor) -> Tensor:
    return ops.geglu_fwd(u)[0]


@geglu.register_fake
def _(u):
    return u.new_empty(u.shape[0], u.shape[1] // 2)


geglu.register_autograd(lambda ctx, dy: torch.ops.neurosis_hip.geglu_bwd(dy.contiguous(), ctx.saved_tensors[0]),
                        setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


# ---------------------------------------------------------------------------------------------------------------
# attention: softmax(q k^T / sqrt(d)) v on token matrices [B*L, heads*d]
# ---------------------------------------------------------------------------------------------------------------
def _attn_desc(q, k, v, o, B, heads):
    d = ops.NkAttnDesc()
    dh = q.shape[1] // heads
    Lq, Lk = q.shape[0] // B, k.shape[0] // B
    d.B, d.H, d.Lq, d.Lk, d.D = B, heads, Lq, Lk, dh
    d.sq, d.sk, d.sv, d.so = q.stride(0), k.stride(0), v.stride(0), o.stride(0)
    d.bq, d.bk, d.bv, d.bo = Lq * q.stride(0), Lk * k.stride(0), Lk * v.stride(0), Lq * o.stride(0)
    d.scale = float(dh) ** -0.5
    d.causal = 0
    return d, Lq, Lk


@_op("attention_fwd")
def attention_fwd(q: Tensor, k: Tensor, v: Tensor, B: int, heads: int) -> Tuple[Tensor, Tensor]:
    import ctypes as C

    o = torch.empty(q.shape[0], q.shape[1], dtype=BF16, device=q.device)
    d, Lq, _ = _attn_desc(q, k, v, o, B, heads)
    lse = torch.empty(B, heads, Lq, dtype=torch.float32, device=q.device)
    ops.call("nk_attention_fwd", C.byref(d), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), ops._stream())
    return o, lse


@attention_fwd.register_fake
def _(q, k, v, B, heads):
    return q.new_empty(q.shape[0], q.shape[1]), q.new_empty(B, heads, q.shape[0] // B, dtype=torch.float32)


@_op("attention_bwd")
def attention_bwd(do: Tensor, q: Tensor, k: Tensor, v: Tensor, o: Tensor, lse: Tensor, B: int, heads: int) -> Tuple[Tensor, Tensor, Tensor]:
    import ctypes as C

    d, Lq, Lk = _attn_desc(q, k, v, o, B, heads)
    dq, dk, dv = (torch.empty(t.shape[0], t.shape[1], dtype=BF16, device=q.device) for t in (q, k, v))
    d.sdq, d.sdk, d.sdv, d.sdo = dq.stride(0), dk.stride(0), dv.stride(0), do.stride(0)
    d.bdq, d.bdk, d.bdv, d.bdo = Lq * dq.stride(0), Lk * dk.stride(0), Lk * dv.stride(0), Lq * do.stride(0)
    ws = ops._ws(ops.query("nk_attention_bwd_ws_floats", C.byref(d)), q.device)
    ops.call("nk_attention_bwd", C.byref(d), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(),
             dv.data_ptr(), ws.data_ptr(), ops._stream())
    return dq, dk, dv


@attention_bwd.register_fake
def _(do, q, k, v, o, lse, B, heads):
    return q.new_empty(q.shape), k.new_empty(k.shape), v.new_empty(v.shape)


@_op("attention")
def attention(q: Tensor, k: Tensor, v: Tensor, B: int, heads: int) -> Tensor:
    return torch.ops.neurosis_hip.attention_fwd(q, k, v, B, heads)[0]


@attention.register_fake
def _(q, k, v, B, heads):
    return q.new_empty(q.shape[0], q.shape[1])


def _attn_setup(ctx, inputs, output):
    q, k, v, B, heads = inputs
    o, lse = torch.ops.neurosis_hip.attention_fwd(q, k, v, B, heads)
    ctx.save_for_backward(q, k, v, o, lse)
    ctx.meta = (B, heads)


def _attn_bwd(ctx, do):
    q, k, v, o, lse = ctx.saved_tensors
    dq, dk, dv = torch.ops.neurosis_hip.attention_bwd(do.contiguous(), q, k, v, o, lse, *ctx.meta)
    return dq, dk, dv, None, None


attention.register_autograd(_attn_bwd, setup_context=_attn_setup)


# ---------------------------------------------------------------------------------------------------------------
# Conv2d as implicit GEMM on channels-last tokens: x [N*H*W, Cin], w [Cout, KH, KW, Cin] (bf16) -> [N*Ho*Wo, Cout]
# ---------------------------------------------------------------------------------------------------------------
def _conv_geom(N, H, W, w, stride, padding):
    Cout, KH, KW, Cin = w.shape
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    return ops._conv_desc(N, H, W, Cin, Cout, KH, KW, stride, padding, padding, Ho, Wo, False), Ho, Wo


@_op("conv2d_fwd")
def conv2d_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], N: int, H: int, W: int, stride: int, padding: int) -> Tensor:
    import ctypes as C

    d, Ho, Wo = _conv_geom(N, H, W, w, stride, padding)
    y = torch.empty(N * Ho * Wo, w.shape[0], dtype=BF16, device=x.device)
    ops.call("nk_conv2d_fwd", C.byref(d), x.data_ptr(), w.data_ptr(), ops._p(bias), None, None, y.data_ptr(), ops._stream())
    return y


@conv2d_fwd.register_fake
def _(x, w, bias, N, H, W, stride, padding):
    KH, KW = w.shape[1], w.shape[2]
    return x.new_empty(N * ((H + 2 * padding - KH) // stride + 1) * ((W + 2 * padding - KW) // stride + 1), w.shape[0])


@_op("conv2d_dgrad")
def conv2d_dgrad(dy: Tensor, w: Tensor, N: int, H: int, W: int, stride: int, padding: int) -> Tensor:
    import ctypes as C

    d, _, _ = _conv_geom(N, H, W, w, stride, padding)
    dx = torch.empty(N * H * W, w.shape[3], dtype=BF16, device=dy.device)
    ops.call("nk_conv2d_dgrad", C.byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), ops._stream())
    return dx


@conv2d_dgrad.register_fake
def _(dy, w, N, H, W, stride, padding):
    return dy.new_empty(N * H * W, w.shape[3])


@_op("conv2d_wgrad")
def conv2d_wgrad(dy: Tensor, x: Tensor, Cout: int, KH: int, KW: int, N: int, H: int, W: int, stride: int, padding: int) -> Tensor:
    import ctypes as C

    Cin = x.shape[1]
    Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
    d = ops._conv_desc(N, H, W, Cin, Cout, KH, KW, stride, padding, padding, Ho, Wo, False)
    dw = torch.empty(Cout, KH, KW, Cin, dtype=torch.float32, device=x.device)
    ops.call("nk_conv2d_wgrad", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 0, ops._stream())
    return dw


@conv2d_wgrad.register_fake
def _(dy, x, Cout, KH, KW, N, H, W, stride, padding):
    return x.new_empty(Cout, KH, KW, x.shape[1], dtype=torch.float32)


@_op("conv2d")
def conv2d(x: Tensor, w: Tensor, bias: Optional[Tensor], N: int, H: int, W: int, stride: int, padding: int) -> Tensor:
    return torch.ops.neurosis_hip.conv2d_fwd(x, w, bias, N, H, W, stride, padding)


@conv2d.register_fake
def _(x, w, bias, N, H, W, stride, padding):
    KH, KW = w.shape[1], w.shape[2]
    return x.new_empty(N * ((H + 2 * padding - KH) // stride + 1) * ((W + 2 * padding - KW) // stride + 1), w.shape[0])


def _conv_setup(ctx, inputs, output):
    x, w, bias, N, H, W, stride, padding = inputs
    ctx.save_for_backward(x, w)
    ctx.meta = (N, H, W, stride, padding, bias is not None)


def _conv_bwd(ctx, dy):
    x, w = ctx.saved_tensors
    N, H, W, stride, padding, has_bias = ctx.meta
    dy = dy.contiguous()
    dx = torch.ops.neurosis_hip.conv2d_dgrad(dy, w, N, H, W, stride, padding) if ctx.needs_input_grad[0] else None
    dw = torch.ops.neurosis_hip.conv2d_wgrad(dy, x, w.shape[0], w.shape[1], w.shape[2], N, H, W, stride, padding).to(w.dtype) if ctx.needs_input_grad[1] else None
    db = torch.ops.neurosis_hip.colsum(dy) if has_bias and ctx.needs_input_grad[2] else None
    return dx, dw, db, None, None, None, None, None


conv2d.register_autograd(_conv_bwd, setup_context=_conv_setup)


# ---------------------------------------------------------------------------------------------------------------
# glue
# ---------------------------------------------------------------------------------------------------------------
@_op("timestep_embedding")
def timestep_embedding(t: Tensor, dim: int, max_period: float = 10000.0) -> Tensor:
    return ops.timestep_embedding(t, dim, max_period)


@timestep_embedding.register_fake
def _(t, dim, max_period=10000.0):
    return t.new_empty(t.shape[0], dim, dtype=BF16)


@_op("nchw_to_nlc")
def nchw_to_nlc(x: Tensor, cpad: int) -> Tensor:
    return ops.nchw_to_tokens(x, cpad)


@nchw_to_nlc.register_fake
def _(x, cpad):
    return x.new_empty(x.shape[0] * x.shape[2] * x.shape[3], cpad, dtype=BF16)


@_op("nlc_to_nchw")
def nlc_to_nchw(t: Tensor, N: int, C: int, H: int, W: int) -> Tensor:
    return ops.tokens_to_nchw(t, N, C, H, W)


@nlc_to_nchw.register_fake
def _(t, N, C, H, W):
    return t.new_empty(N, C, H, W, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------------------------
# upsample2x_nearest_cat: the two data-movement ops of the UNet decoder (openaimodel.py:140 F.interpolate(scale_factor=2, mode="nearest")
# in Upsample.forward -- here folded into the following 3 x 3 convolution's gather, so only its BACKWARD exists as a kernel -- and :836
# torch.cat([h, hs.pop()], dim=1) on channels-last tokens).
# ---------------------------------------------------------------------------------------------------------------
@_op("cat_channels")
def cat_channels(a: Tensor, b: Tensor) -> Tensor:
    out = torch.empty(a.shape[0], a.shape[1] + b.shape[1], dtype=BF16, device=a.device)
    ops.call("nk_cat_channels", a.data_ptr(), b.data_ptr(), out.data_ptr(), a.shape[0], a.shape[1], b.shape[1], ops._stream())
    return out


@cat_channels.register_fake
def _(a, b):
    return a.new_empty(a.shape[0], a.shape[1] + b.shape[1])


@_op("split_channels")
def split_channels(x: Tensor, Ca: int) -> Tuple[Tensor, Tensor]:
    rows, Cb = x.shape[0], x.shape[1] - Ca
    a, b = torch.empty(rows, Ca, dtype=BF16, device=x.device), torch.empty(rows, Cb, dtype=BF16, device=x.device)
    ops.call("nk_split_channels", x.data_ptr(), a.data_ptr(), b.data_ptr(), rows, Ca, Cb, ops._stream())
    return a, b


@split_channels.register_fake
def _(x, Ca):
    return x.new_empty(x.shape[0], Ca), x.new_empty(x.shape[0], x.shape[1] - Ca)


def _cat_setup(ctx, inputs, output):
    ctx.Ca = inputs[0].shape[1]


def _cat_bwd(ctx, dout):
    da, db = torch.ops.neurosis_hip.split_channels(dout.contiguous(), ctx.Ca)
    return da, db


cat_channels.register_autograd(_cat_bwd, setup_context=_cat_setup)


@_op("upsample2x_nearest_bwd")
def upsample2x_nearest_bwd(dup: Tensor, N: int, H: int, W: int) -> Tensor:
    """dup [N*2H*2W, C] (gradient on the upsampled grid) -> [N*H*W, C]: the 2 x 2 sum-pool that is nearest-upsampling's adjoint"""
    C_ = dup.shape[1]
    dx = torch.empty(N * H * W, C_, dtype=BF16, device=dup.device)
    ops.call("nk_upsample2x_bwd", dup.data_ptr(), dx.data_ptr(), N, H, W, C_, ops._stream())
    return dx


@upsample2x_nearest_bwd.register_fake
def _(dup, N, H, W):
    return dup.new_empty(N * H * W, dup.shape[1])


@_op("upsample2x_nearest_conv")
def upsample2x_nearest_conv(x: Tensor, w: Tensor, bias: Optional[Tensor], N: int, H: int, W: int) -> Tensor:
    """Upsample.forward with use_conv (openaimodel.py:126-143): nearest x2 then 3 x 3 / padding 1 conv, the upsampled map never written"""
    import ctypes as C

    Cout, KH, KW, Cin = w.shape
    d = ops._conv_desc(N, H, W, Cin, Cout, KH, KW, 1, 1, 1, 2 * H, 2 * W, True)
    y = torch.empty(N * 4 * H * W, Cout, dtype=BF16, device=x.device)
    ops.call("nk_conv2d_fwd", C.byref(d), x.data_ptr(), w.data_ptr(), ops._p(bias), None, None, y.data_ptr(), ops._stream())
    return y


@upsample2x_nearest_conv.register_fake
def _(x, w, bias, N, H, W):
    return x.new_empty(N * 4 * H * W, w.shape[0])


# ---------------------------------------------------------------------------------------------------------------
# edm_loss: StandardDiffusionLoss "edm" arithmetic around the network (loss.py:117-157, denoiser.py:41-53, functions.py:91-94)
# ---------------------------------------------------------------------------------------------------------------
@_op("edm_prepare")
def edm_prepare(x: Tensor, eps: Tensor, sigma: Tensor, c_in: Tensor, cpad: int) -> Tuple[Tensor, Tensor]:
    """z_t = x + sigma * eps (fp32 NCHW) and the network input bf16(z_t * c_in) as channels-last tokens padded to cpad channels"""
    B, Cc, H, W = x.shape
    zt = torch.empty_like(x)
    net_in = torch.empty(B * H * W, cpad, dtype=BF16, device=x.device)
    ops.call("nk_edm_prepare", x.data_ptr(), eps.data_ptr(), sigma.data_ptr(), c_in.data_ptr(), zt.data_ptr(), net_in.data_ptr(), B, Cc, H * W, cpad, ops._stream())
    return zt, net_in


@edm_prepare.register_fake
def _(x, eps, sigma, c_in, cpad):
    return torch.empty_like(x), x.new_empty(x.shape[0] * x.shape[2] * x.shape[3], cpad, dtype=BF16)


@_op("edm_loss_fwd")
def edm_loss_fwd(net_out: Tensor, zt: Tensor, target: Tensor, c_out: Tensor, c_skip: Tensor, w: Tensor) -> Tensor:
    """loss[b] = w[b] * mean((net_out * c_out + z_t * c_skip - target)^2): net_out bf16 tokens [B*H*W, Cpad], z_t / target fp32 NCHW"""
    B, Cc, H, W = zt.shape
    loss = torch.empty(B, dtype=torch.float32, device=zt.device)
    ops.call("nk_edm_loss", net_out.data_ptr(), zt.data_ptr(), target.data_ptr(), c_out.data_ptr(), c_skip.data_ptr(), w.data_ptr(), loss.data_ptr(), None,
             B, Cc, H * W, net_out.shape[1], 1.0, ops._stream())
    return loss


@edm_loss_fwd.register_fake
def _(net_out, zt, target, c_out, c_skip, w):
    return zt.new_empty(zt.shape[0], dtype=torch.float32)


@_op("edm_loss_bwd")
def edm_loss_bwd(dloss: Tensor, net_out: Tensor, zt: Tensor, target: Tensor, c_out: Tensor, c_skip: Tensor, w: Tensor) -> Tensor:
    """d sum_b(dloss[b] * loss[b]) / d net_out, bf16 tokens"""
    B, Cc, H, W = zt.shape
    wg = (w * dloss.float()).contiguous()
    dnet = torch.empty_like(net_out)
    scratch = torch.empty(B, dtype=torch.float32, device=zt.device)
    ops.call("nk_edm_loss", net_out.data_ptr(), zt.data_ptr(), target.data_ptr(), c_out.data_ptr(), c_skip.data_ptr(), wg.data_ptr(), scratch.data_ptr(),
             dnet.data_ptr(), B, Cc, H * W, net_out.shape[1], 1.0, ops._stream())
    return dnet


@edm_loss_bwd.register_fake
def _(dloss, net_out, zt, target, c_out, c_skip, w):
    return torch.empty_like(net_out)


@_op("edm_loss")
def edm_loss(net_out: Tensor, zt: Tensor, target: Tensor, c_out: Tensor, c_skip: Tensor, w: Tensor) -> Tensor:
    return torch.ops.neurosis_hip.edm_loss_fwd(net_out, zt, target, c_out, c_skip, w)


@edm_loss.register_fake
def _(net_out, zt, target, c_out, c_skip, w):
    return zt.new_empty(zt.shape[0], dtype=torch.float32)


def _edm_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _edm_bwd(ctx, dloss):
    net_out, zt, target, c_out, c_skip, w = ctx.saved_tensors
    return torch.ops.neurosis_hip.edm_loss_bwd(dloss.contiguous(), net_out, zt, target, c_out, c_skip, w), None, None, None, None, None


edm_loss.register_autograd(_edm_bwd, setup_context=_edm_setup)


# ---------------------------------------------------------------------------------------------------------------
# flat_allreduce_{start, wait}: the data-parallel exchange of a flat gradient buffer as two ops (what Lightning's DDP reducer does
# behind autograd hooks for the reference; neurosis_amd.dp.FlatGradReducer underneath: RCCL on its own stream, joined by `wait`).
# The reducer is found by the buffer's address: one per flat gradient buffer and process.
# ---------------------------------------------------------------------------------------------------------------
_reducers: dict = {}


def _reducer_for(flat: Tensor):
    from .dp import FlatGradReducer

    key = (flat.data_ptr(), flat.numel())
    r = _reducers.get(key)
    if r is None:
        r = _reducers[key] = FlatGradReducer(flat)
    return r


@_op("flat_allreduce_start", mutates=("flat",))
def flat_allreduce_start(flat: Tensor, lo: int, hi: int) -> None:
    """begin summing flat[lo:hi] over the ranks (asynchronous: ordered behind the current stream, running on the exchange stream)"""
    _reducer_for(flat).reduce_range(lo, hi)


@flat_allreduce_start.register_fake
def _(flat, lo, hi):
    return None


@_op("flat_allreduce_wait", mutates=("flat",))
def flat_allreduce_wait(flat: Tensor) -> None:
    """make the current stream wait for every reduction started on this buffer"""
    _reducer_for(flat).finish()


@flat_allreduce_wait.register_fake
def _(flat):
    return None


# ---------------------------------------------------------------------------------------------------------------
# the fused entry points of round 3
# ---------------------------------------------------------------------------------------------------------------
@_op("conv2d_fwd_stats")
def conv2d_fwd_stats(x: Tensor, w: Tensor, bias: Optional[Tensor], N: int, H: int, W: int, groups: int) -> Tuple[Tensor, Tensor]:
    """3 x 3 / stride 1 / padding 1 convolution that also returns the GroupNorm sums of its output, [N, 2 * groups] fp32
    (entry 2g = sum, 2g + 1 = sum of squares); from the kernel's epilogue where the halo-tile kernel takes the shape, else a statistics pass"""
    import ctypes as C

    Cout, KH, KW, Cin = w.shape
    d = ops._conv_desc(N, H, W, Cin, Cout, KH, KW, 1, 1, 1, H, W, False)
    y = torch.empty(N * H * W, Cout, dtype=BF16, device=x.device)
    sums = torch.empty(N, 2 * groups, dtype=torch.float32, device=x.device)
    tiles = ops.query("nk_conv2d_stats_tiles", C.byref(d), groups) if (KH, KW) == (3, 3) else 0
    if tiles:
        part = torch.empty(N, tiles, 2 * groups, dtype=torch.float32, device=x.device)
        ops.call("nk_conv2d_fwd_stats", C.byref(d), x.data_ptr(), w.data_ptr(), ops._p(bias), None, None, y.data_ptr(), part.data_ptr(), groups, ops._stream())
        ws = ops._ws(ops.query("nk_groupnorm_sums_ws_floats", N, tiles, groups), x.device)
        ops.call("nk_groupnorm_sums_from_parts", part.data_ptr(), sums.data_ptr(), ws.data_ptr(), N, tiles, groups, ops._stream())
    else:
        ops.call("nk_conv2d_fwd", C.byref(d), x.data_ptr(), w.data_ptr(), ops._p(bias), None, None, y.data_ptr(), ops._stream())
        ws = ops._ws(ops.query("nk_groupnorm_ws_floats", N, H * W, Cout, groups), x.device)
        ops.call("nk_groupnorm_sums", y.data_ptr(), sums.data_ptr(), ws.data_ptr(), N, H * W, Cout, groups, ops._stream())
    return y, sums


@conv2d_fwd_stats.register_fake
def _(x, w, bias, N, H, W, groups):
    return x.new_empty(N * H * W, w.shape[0]), x.new_empty(N, 2 * groups, dtype=torch.float32)


@_op("linear_fwd_geglu")
def linear_fwd_geglu(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """FeedForward.net[0] (GEGLU): u = x @ w^T + bias [M, 2I] and h = u[:, :I] * gelu(u[:, I:]) [M, I] -- one launch where the 256 x 256 kernel
    takes the shape (ops.linear_geglu_fwd), else the projection followed by the GEGLU kernel"""
    M, K = x.shape
    I = w.shape[0] // 2
    if x.is_contiguous() and w.is_contiguous() and ops.query("nk_linear_fwd_geglu_ok", M, I, K):
        u = torch.empty(M, 2 * I, dtype=BF16, device=x.device)
        h = torch.empty(M, I, dtype=BF16, device=x.device)
        ops.call("nk_linear_fwd_geglu", x.data_ptr(), w.data_ptr(), ops._p(bias), u.data_ptr(), h.data_ptr(), M, I, K, x.stride(0), w.stride(0), u.stride(0),
                 h.stride(0), ops._stream())
        return u, h
    u = torch.ops.neurosis_hip.linear_fwd(x, w, bias)
    return u, torch.ops.neurosis_hip.geglu_fwd(u)


@linear_fwd_geglu.register_fake
def _(x, w, bias):
    return x.new_empty(x.shape[0], w.shape[0]), x.new_empty(x.shape[0], w.shape[0] // 2)


@_op("linear_dgrad_geglu")
def linear_dgrad_geglu(dy: Tensor, w: Tensor, u: Tensor) -> Tensor:
    """FeedForward backward through net[2] and the GEGLU in one launch: du [M, 2I] from dy [M, N], w [N, I], u = [a | g] [M, 2I]"""
    M, N = dy.shape
    I = w.shape[1]
    du = torch.empty(M, 2 * I, dtype=BF16, device=dy.device)
    ops.call("nk_linear_dgrad_geglu", dy.data_ptr(), w.data_ptr(), u.data_ptr(), du.data_ptr(), M, N, I, dy.stride(0), w.stride(0), u.stride(0), du.stride(0),
             ops._stream())
    return du


@linear_dgrad_geglu.register_fake
def _(dy, w, u):
    return dy.new_empty(dy.shape[0], 2 * w.shape[1])


OPS = ("cat_channels", "split_channels", "upsample2x_nearest_bwd", "upsample2x_nearest_conv", "edm_prepare", "edm_loss_fwd", "edm_loss_bwd", "edm_loss",
       "flat_allreduce_start", "flat_allreduce_wait", "conv2d_fwd_stats", "linear_dgrad_geglu", "linear_fwd_geglu", "linear_fwd", "linear_dgrad", "linear_wgrad", "colsum", "linear", "layernorm_fwd", "layernorm_bwd", "layernorm", "groupnorm_silu_fwd",
       "groupnorm_silu_bwd", "groupnorm_silu", "geglu_fwd", "geglu_bwd", "geglu", "attention_fwd", "attention_bwd", "attention", "conv2d_fwd",
       "conv2d_dgrad", "conv2d_wgrad", "conv2d", "timestep_embedding", "nchw_to_nlc", "nlc_to_nchw")
