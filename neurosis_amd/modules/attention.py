"""Transformer blocks of the UNet on MI355X: the class surface of neurosis.modules.attention
(/root/reference/src/neurosis/modules/attention.py) over the HIP kernels.

Same constructor arguments, attribute names and state_dict keys as the reference classes; the arithmetic
(projections, flash attention, LayerNorm, GEGLU) runs in libneurosis_hip.so on bf16 token matrices.
Every reference attention backend name ("softmax", "softmax-xformers", "torch-sdp") selects the same HIP
flash-attention kernel: they are one mathematical function (attention.py:187-417).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor, nn

from .. import ops
from ..nn import adjacent, apply_module, as_tokens, linear_module_fwd
from ..ops import BF16, Img


def zero_module(module: nn.Module) -> nn.Module:
    """modules/diffusion/util.py:180-186."""
    for p in module.parameters():
        p.detach().zero_()
    return module


class GEGLU(nn.Module):
    """attention.py:50-57."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def fwd(self, x: Tensor):
        u, b_proj = linear_module_fwd(self.proj, x)
        y, b_g = ops.geglu_fwd(u)

        def bwd(dy: Tensor):
            return b_proj(b_g(dy))

        return y, bwd

    def forward(self, x: Tensor) -> Tensor:
        return _token_module_forward(self, x)


class FeedForward(nn.Module):
    """attention.py:60-74 (glu=True is what BasicTransformerBlock builds; the GELU variant is kept for parity)."""

    def __init__(self, dim: int, dim_out: Optional[int] = None, mult: int = 4, glu: bool = False, dropout: float = 0.0):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("dropout > 0 is not on the SDXL training path (configs use 0.0)")
        inner_dim = int(dim * mult)
        dim_out = dim_out or dim
        if not glu:
            raise NotImplementedError("FeedForward(glu=False) is unused by SD1.5/SDXL UNets")
        self.net = nn.Sequential(GEGLU(dim, inner_dim), nn.Dropout(dropout), nn.Linear(inner_dim, dim_out))

    def fwd(self, x: Tensor, residual: Optional[Tensor] = None, x_saved=None, recompute_h: bool = False):
        """x_saved / recompute_h (selective recompute, BasicTransformerBlock.recompute): the projection's input is rebuilt by `x_saved()` and
        the GEGLU product h = a * gelu(g) from the kept projection output u when the weight gradients need them; neither is held."""
        proj = self.net[0].proj
        # round 6: unless h must be rebuilt from u in backward (selective recompute), the projection's epilogue keeps the saved-derivative
        # form s = [gelu(g) | a gelu'(g)] instead of u = [a | g]: the backward's epilogue is then two products per element
        save_s = not recompute_h and ops.geglu_save_enabled()
        u, g, b_proj = ops.linear_geglu_fwd(x, proj.weight, proj.bias, x_saved, save_derivative=save_s)     # the GEGLU rides in the projection's epilogue where it can
        y, b_out = linear_module_fwd(self.net[2], g, residual, x_saved=(lambda: ops.geglu_fwd(u)[0]) if recompute_h else None)
        del g

        def bwd(dy: Tensor):
            # net[2]'s input gradient with the GEGLU backward in its epilogue: one launch, d(a * gelu(g)) never goes to HBM
            return b_proj(b_out(dy, geglu_s=u) if save_s else b_out(dy, geglu_u=u))

        return y, bwd

    def forward(self, x: Tensor) -> Tensor:
        return _token_module_forward(self, x)


def _token_module_forward(mod: nn.Module, x: Tensor) -> Tensor:
    shape = x.shape

    def run(x):
        y, bwd = mod.fwd(as_tokens(x))
        return y.view(*shape[:-1], y.shape[-1]), lambda g: bwd(as_tokens(g)).view(shape)

    return apply_module(run, [x], mod)


class CrossAttention(nn.Module):
    """attention.py:187-417 (CrossAttention / MemoryEfficientCrossAttention / TorchSDPCrossAttention share this
    parameter layout: to_q, to_k, to_v without bias, to_out = [Linear, Dropout])."""

    def __init__(self, query_dim: int, context_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64, dropout: float = 0.0, backend=None, **kwargs):
        super().__init__()
        if dropout != 0.0:
            raise NotImplementedError("attention dropout > 0 is not on the SDXL training path")
        inner_dim = dim_head * heads
        context_dim = context_dim or query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    def fwd(self, x: Tensor, context: Optional[Tensor], B: int, residual: Optional[Tensor] = None, need_dctx: bool = False, x_saved=None):
        """x [B*L, C]; context [B*Lc, Cc] or None (self-attention).  Returns (out, bwd); bwd(dy) -> (dx, dctx|None).
        x_saved: a callable that rebuilds x for the weight gradients (selective recompute: the closures then do not hold x)."""
        wq, wk, wv = self.to_q.weight, self.to_k.weight, self.to_v.weight
        x_back = (lambda: x_keep) if x_saved is None else x_saved
        x_keep = x if x_saved is None else None
        inner = wq.shape[0]
        self_attn = context is None
        acc = lambda: ops.wgrad_mode(wq)
        if self_attn and adjacent(wq, wk, wv):
            # one projection GEMM: [to_q; to_k; to_v] are back to back in the flat parameter store
            w_qkv = torch.as_strided(ops.shadow(wq), (3 * inner, wq.shape[1]), (wq.shape[1], 1))
            qkv = ops.gemm_nt(x, w_qkv)
            q, k, v = qkv[:, :inner], qkv[:, inner:2 * inner], qkv[:, 2 * inner:]
            o, b_att = ops.attention_fwd(q, k, v, B, self.heads, self.dim_head)
            y, b_out = linear_module_fwd(self.to_out[0], o, residual)

            def bwd(dy: Tensor):
                do = b_out(dy)
                dqkv = torch.empty_like(qkv)
                b_att(do, dqkv[:, :inner], dqkv[:, inner:2 * inner], dqkv[:, 2 * inner:])
                g_qkv = torch.as_strided(ops.grad_flat(wq), (3 * inner, wq.shape[1]), (wq.shape[1], 1))
                xb = x_back()
                ops.on_wgrad_stream(lambda: ops.gemm_tn_f32(dqkv, xb, g_qkv, acc()), dqkv, xb, owner=wq)
                return ops.gemm_nn(dqkv, w_qkv), None

            del x
            return y, bwd

        ctx = x if self_attn else context
        ctx_back = x_back if self_attn else (lambda: context)       # what the K / V weight gradients read, rebuilt with x when it is x
        q = ops.gemm_nt(x, ops.w2d(wq))
        fused_kv = adjacent(wk, wv)
        if fused_kv:
            w_kv = torch.as_strided(ops.shadow(wk), (2 * inner, wk.shape[1]), (wk.shape[1], 1))
            # UNetModel.fwd projects the (block-independent) context for every cross-attention up front, eight blocks per launch,
            # and leaves the result here; anything else (a lone module, a checkpointed re-run) projects it now
            pre = self.__dict__.pop("_nk_kv", None) if not self_attn else None
            kv = pre[0] if pre is not None and pre[1] is ctx else None       # (only the projection of THIS context object)
            if kv is None:
                kv = ops.gemm_nt(ctx, w_kv)
            k, v = kv[:, :inner], kv[:, inner:]
        else:
            k = ops.gemm_nt(ctx, ops.w2d(wk))
            v = ops.gemm_nt(ctx, ops.w2d(wv))
        o, b_att = ops.attention_fwd(q, k, v, B, self.heads, self.dim_head)
        y, b_out = linear_module_fwd(self.to_out[0], o, residual)
        del x, ctx                      # (the closure below reaches the layer's input through x_back / ctx_back only)

        def bwd(dy: Tensor):
            do = b_out(dy)
            xb = x_back()
            cb = xb if self_attn else ctx_back()
            if fused_kv:
                dkv = torch.empty_like(kv)
                dq, _, _ = b_att(do, None, dkv[:, :inner], dkv[:, inner:])
                g_kv = torch.as_strided(ops.grad_flat(wk), (2 * inner, wk.shape[1]), (wk.shape[1], 1))
                ops.on_wgrad_stream(lambda: ops.gemm_tn_f32(dkv, cb, g_kv, acc()), dkv, cb, owner=wq)
            else:
                dq, dk, dv = b_att(do)

                def wg_kv():
                    ops.gemm_tn_f32(dk, cb, ops.g2d(wk), acc())
                    ops.gemm_tn_f32(dv, cb, ops.g2d(wv), acc())

                ops.on_wgrad_stream(wg_kv, dk, dv, cb, owner=wq)
            if ops._wgrad_queue is not None and ops._wgrad_queue.takes(wq) and dq.is_contiguous() and xb.is_contiguous():
                ops._wgrad_queue.add(dq, xb, ops.g2d(wq))
            else:
                ops.on_wgrad_stream(lambda: ops.gemm_tn_f32(dq, xb, ops.g2d(wq), acc()), dq, xb, owner=wq)
            dx = ops.gemm_nn(dq, ops.w2d(wq))
            dctx = None
            if self_attn or need_dctx:
                if fused_kv:
                    dc = ops.gemm_nn(dkv, w_kv, dx if self_attn else None)
                else:
                    dc = ops.gemm_nn(dk, ops.w2d(wk), dx if self_attn else None)
                    dc = ops.gemm_nn(dv, ops.w2d(wv), dc)
                if self_attn:
                    dx = dc
                else:
                    dctx = dc
            return dx, dctx

        return y, bwd

    def forward(self, x: Tensor, context: Optional[Tensor] = None, mask=None, additional_tokens=None, n_times_crossframe_attn_in_self: int = 0) -> Tensor:
        if mask is not None or additional_tokens is not None or n_times_crossframe_attn_in_self:
            raise NotImplementedError("mask / additional_tokens / cross-frame attention are video features outside the SDXL path")
        B, L, _ = x.shape
        ins = [x] if context is None else [x, context]

        def run(x, context=None):
            need_dctx = context is not None and context.requires_grad
            y, bwd = self.fwd(as_tokens(x), None if context is None else as_tokens(context), B, need_dctx=need_dctx)

            def bwd2(g):
                dx, dctx = bwd(as_tokens(g))
                dx = dx.view(x.shape)
                if context is None:
                    return dx
                return dx, (None if dctx is None else dctx.view(context.shape))

            return y.view(B, L, -1), bwd2

        return apply_module(run, ins, self)


# the reference's three backends are one function here
MemoryEfficientCrossAttention = CrossAttention
TorchSDPCrossAttention = CrossAttention


class BasicTransformerBlock(nn.Module):
    """attention.py:420-511."""

    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention, "torch-sdp": TorchSDPCrossAttention}

    def __init__(self, dim: int, n_heads: int, d_head: int, dropout: float = 0.0, context_dim: Optional[int] = None, gated_ff: bool = True,
                 checkpoint: bool = True, disable_self_attn: bool = False, attn_mode: str = "softmax", sdp_backend=None):
        super().__init__()
        if attn_mode not in self.ATTENTION_MODES:
            raise ValueError(f"Unknown attention mode: {attn_mode}")
        attn_cls = self.ATTENTION_MODES[attn_mode]
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout, context_dim=context_dim if disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint
        # Selective recompute (the reference only has the whole-block torch.utils.checkpoint above): "norms" keeps
        # every GEMM output (the fused q/k/v, the attention outputs, the FeedForward projection u) and REBUILDS in backward what is cheap to
        # rebuild -- the three LayerNorm outputs (from the residual stream, which the LayerNorm backward holds anyway) and the GEGLU
        # product a * gelu(g) (from u) -- instead of holding them: 7 of the block's 19 saved token matrices' worth of bytes for two
        # elementwise passes.  Bit-identical results (the same kernels on the same inputs).  UNetModel.set_recompute() sets it network-wide.
        self.recompute: Optional[str] = None

    def _fwd(self, x: Tensor, context: Optional[Tensor], B: int, need_dctx: bool):
        if self.recompute not in (None, "norms"):
            raise ValueError(f"BasicTransformerBlock.recompute must be None or 'norms', got {self.recompute!r}")
        lean = self.recompute == "norms" and ops.recording()
        again = (lambda t, n: (lambda: ops.layernorm_fwd(t, n.weight, n.bias, n.eps)[0])) if lean else (lambda t, n: None)
        n1, b_n1 = ops.layernorm_fwd(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        a1, b_a1 = self.attn1.fwd(n1, context if self.disable_self_attn else None, B, residual=x, need_dctx=need_dctx, x_saved=again(x, self.norm1))
        del n1
        n2, b_n2 = ops.layernorm_fwd(a1, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        a2, b_a2 = self.attn2.fwd(n2, context, B, residual=a1, need_dctx=need_dctx, x_saved=again(a1, self.norm2))
        del n2
        n3, b_n3 = ops.layernorm_fwd(a2, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        y, b_ff = self.ff.fwd(n3, residual=a2, x_saved=again(a2, self.norm3), recompute_h=lean)
        del n3

        def bwd(dy: Tensor):
            with ops.batched_wgrads(self.norm1.weight):   # the block's small same-shape weight gradients go out as one launch
                da2 = b_n3(b_ff(dy), dy)       # LN3 backward + the residual branch of x + ff(...)
                dn2, dctx2 = b_a2(da2)
                da1 = b_n2(dn2, da2)
                dn1, dctx1 = b_a1(da1)
                dx = b_n1(dn1, da1)
            dctx = dctx2
            if dctx1 is not None:
                dctx = dctx1 if dctx is None else ops.add(dctx, dctx1)
            return dx, dctx

        return y, bwd

    def fwd(self, x: Tensor, context: Optional[Tensor], B: int, need_dctx: bool = False):
        """x [B*L, C] tokens.  With `checkpoint` the block's activations are dropped and recomputed in backward
        (the reference wraps _forward in torch.utils.checkpoint, attention.py:482-485)."""
        if not (self.checkpoint and ops.recording()):
            return self._fwd(x, context, B, need_dctx)
        y, _ = self._fwd(x, context, B, need_dctx)

        def bwd(dy: Tensor):
            _, b = self._fwd(x, context, B, need_dctx)
            return b(dy)

        return y, bwd

    def forward(self, x: Tensor, context: Optional[Tensor] = None, additional_tokens=None, n_times_crossframe_attn_in_self: int = 0) -> Tensor:
        if additional_tokens is not None or n_times_crossframe_attn_in_self:
            raise NotImplementedError("video-only arguments are outside the SDXL path")
        B, L, _ = x.shape
        ins = [x] if context is None else [x, context]

        def run(x, context=None):
            need_dctx = context is not None and context.requires_grad
            y, bwd = self.fwd(as_tokens(x), None if context is None else as_tokens(context), B, need_dctx)

            def bwd2(g):
                dx, dctx = bwd(as_tokens(g))
                if context is None:
                    return dx.view(x.shape)
                return dx.view(x.shape), (None if dctx is None else dctx.view(context.shape))

            return y.view(x.shape), bwd2

        return apply_module(run, ins, self)


class SpatialTransformer(nn.Module):
    """attention.py:567-667.  On channels-last data "b c h w -> b (h w) c" is the identity, and the 1x1-conv
    proj_in/proj_out of use_linear=False is the same contraction as the Linear of use_linear=True."""

    def __init__(self, in_channels: int, n_heads: int, d_head: int, depth: int = 1, dropout: float = 0.0, context_dim=None,
                 disable_self_attn: bool = False, use_linear: bool = False, attn_type: str = "softmax", use_checkpoint: bool = True, sdp_backend=None):
        super().__init__()
        if context_dim is not None:
            if not isinstance(context_dim, list):
                context_dim = [context_dim]
            if len(context_dim) != depth:
                if not all(c == context_dim[0] for c in context_dim):
                    raise ValueError("need homogenous context_dim to match depth automatically")
                context_dim = [context_dim[0]] * depth
        else:
            context_dim = [None] * depth
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)
        inner_dim = n_heads * d_head
        if not use_linear:
            from ..nn import Conv2d

            self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        else:
            self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d], disable_self_attn=disable_self_attn,
                                   attn_mode=attn_type, checkpoint=use_checkpoint) for d in range(depth)]
        )
        if not use_linear:
            from ..nn import Conv2d

            self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        else:
            self.proj_out = zero_module(nn.Linear(inner_dim, in_channels))
        self.use_linear = use_linear

    def fwd(self, x: Img, context: Optional[Tensor], need_dctx: bool = False):
        """x Img; context dense tokens [B*Lc, Cc].  bwd(dy tokens) -> (dx tokens, dctx|None)."""
        xn, b_gn = ops.groupnorm_fwd(x, self.norm.weight, self.norm.bias, 32, self.norm.eps, silu=False)
        h, b_in = ops.linear_fwd(xn.t, self.proj_in.weight, self.proj_in.bias)
        blocks = []
        for blk in self.transformer_blocks:
            h, b = blk.fwd(h, context, x.N, need_dctx)
            blocks.append(b)
        y, b_out = ops.linear_fwd(h, self.proj_out.weight, self.proj_out.bias, residual=x.t)

        def bwd(dy: Tensor):
            dh = b_out(dy)
            dctx = None
            for b in reversed(blocks):
                dh, dc = b(dh)
                if dc is not None:
                    dctx = dc if dctx is None else ops.add(dctx, dc)
            blocks.clear()
            dx = b_gn(b_in(dh), dy)   # GroupNorm backward + the "+ x_in" branch
            return dx, dctx

        return Img(y, x.N, x.H, x.W), bwd

    def forward(self, x: Tensor, context: Optional[Tensor] = None) -> Tensor:
        if isinstance(context, list):
            if len(context) != 1:
                raise NotImplementedError("per-block context lists are not used by the SDXL configs")
            context = context[0]
        ins = [x] if context is None else [x, context]

        def run(x, context=None):
            need_dctx = context is not None and context.requires_grad
            img = Img.from_nchw(x)
            out, bwd = self.fwd(img, None if context is None else as_tokens(context), need_dctx)

            def bwd2(g):
                dx, dctx = bwd(Img.from_nchw(g).t)
                dxi = Img(dx, img.N, img.H, img.W).to_nchw()
                if context is None:
                    return dxi
                return dxi, (None if dctx is None else dctx.view(context.shape))

            return out.to_nchw(), bwd2

        return apply_module(run, ins, self)
