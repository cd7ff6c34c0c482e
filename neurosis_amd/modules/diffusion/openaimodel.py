"""UNetModel and its blocks on MI355X: the class surface of neurosis.modules.diffusion.openaimodel
(/root/reference/src/neurosis/modules/diffusion/openaimodel.py) over the HIP kernels.

Constructor arguments, attribute names, module tree and state_dict keys match the reference, so a neurosis
YAML config or checkpoint selects this implementation by class path alone.  Internally the whole UNet runs
on channels-last bf16 token matrices: forward and backward are explicit chains of HIP kernel launches (one
autograd node for the whole network), not a PyTorch op graph.
"""
from __future__ import annotations

import os

from typing import Callable, List, Optional, Union

import torch
from torch import Tensor, nn

from ... import ops
from ...nn import Conv2d, FlatParamStore, apply_module, as_tokens, linear_module_fwd
from ...ops import BF16, Img
from ..attention import SpatialTransformer, zero_module


def conv_nd(dims: int, *args, **kwargs) -> nn.Module:
    """modules/diffusion/util.py:206-218 -- only dims=2 is on the SD/SDXL path."""
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the MI355X path implements 2-D convolutions)")
    return Conv2d(*args, **kwargs)


class TimestepBlock(nn.Module):
    """openaimodel.py:52-62."""

    def forward(self, x: Tensor, emb: Tensor):
        raise NotImplementedError("TimestepBlock is an interface: subclasses implement forward(x, emb)")


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """openaimodel.py:65-93: passes emb to TimestepBlocks, context to SpatialTransformers."""

    def fwd(self, x: Img, emb: Tensor, context: Optional[Tensor]):
        """bwd(dy tokens) -> (dx tokens | None, demb | None)."""
        bwds = []
        first = True
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x, b = layer.fwd(x, emb)
                bwds.append(("emb", b))
            elif isinstance(layer, SpatialTransformer):
                x, b = layer.fwd(x, context)
                bwds.append(("ctx", b))
            elif isinstance(layer, (Conv2d, Upsample, Downsample)):
                x, b = layer.fwd(x, need_dx=not (first and getattr(self, "_nk_input_block", False)))
                bwds.append(("plain", b))
            elif isinstance(layer, nn.Identity):
                continue
            else:
                raise NotImplementedError(f"TimestepEmbedSequential: unsupported layer {type(layer).__name__}")
            first = False

        def bwd(dy: Tensor):
            demb = None
            for kind, b in reversed(bwds):
                if dy is None:
                    break
                if kind == "emb":
                    dy, de = b(dy)
                    demb = de if demb is None else ops.add(demb, de)
                elif kind == "ctx":
                    dy, _ = b(dy)
                else:
                    dyi, _ = b(dy)
                    dy = None if dyi is None else dyi.t
            bwds.clear()
            return dy, demb

        return x, bwd

    def forward(self, x: Tensor, emb: Tensor, context: Optional[Tensor] = None, image_only_indicator=None, time_context=None, num_video_frames=None):
        ins = [x, emb] if context is None else [x, emb, context]

        def run(x, emb, context=None):
            img = Img.from_nchw(x)
            out, bwd = self.fwd(img, as_tokens(emb), None if context is None else as_tokens(context))

            def bwd2(g):
                dx, demb = bwd(Img.from_nchw(g).t)
                res = (Img(dx, img.N, img.H, img.W).to_nchw(), None if demb is None else demb.view(emb.shape))
                return res if context is None else (*res, None)

            return out.to_nchw(), bwd2

        return apply_module(run, ins, self)


class Upsample(nn.Module):
    """openaimodel.py:96-143: nearest 2x (fused into the conv's gather) then 3x3 conv."""

    def __init__(self, channels: int, use_conv: bool, dims: int = 2, out_channels: Optional[int] = None, padding: int = 1,
                 third_up: bool = False, kernel_size: int = 3, scale_factor: int = 2):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2 or scale_factor != 2 or not use_conv:
            raise NotImplementedError("Upsample: the SD/SDXL path uses dims=2, scale 2, conv_resample=True")
        self.conv = conv_nd(dims, self.channels, self.out_channels, kernel_size, padding=padding)

    def fwd(self, x: Img, need_dx: bool = True):
        assert x.C == self.channels
        return self.conv.fwd(x, upsample=True, need_dx=need_dx)

    def forward(self, x: Tensor) -> Tensor:
        return _plain_forward(self, x)


class Downsample(nn.Module):
    """openaimodel.py:146-197: 3x3 stride-2 conv."""

    def __init__(self, channels: int, use_conv: bool, dims: int = 2, out_channels: Optional[int] = None, padding: int = 1, third_down: bool = False):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2 or not use_conv:
            raise NotImplementedError("Downsample: the SD/SDXL path uses dims=2, conv_resample=True")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def fwd(self, x: Img, need_dx: bool = True):
        assert x.C == self.channels
        return self.op.fwd(x, need_dx=need_dx)

    def forward(self, x: Tensor) -> Tensor:
        return _plain_forward(self, x)


def _plain_forward(mod: nn.Module, x: Tensor) -> Tensor:
    def run(x):
        img = Img.from_nchw(x)
        out, bwd = mod.fwd(img)
        return out.to_nchw(), lambda g: bwd(Img.from_nchw(g).t)[0].to_nchw()

    return apply_module(run, [x], mod)


class ResBlock(TimestepBlock):
    """openaimodel.py:200-342.  h = conv(silu(GN(x))) + emb ; out = skip(x) + conv(silu(GN(h))).
    GN+SiLU is one kernel, the emb add and the skip add are conv epilogues, the skip-path gradient is an
    epilogue of the first GroupNorm's backward."""

    def __init__(self, channels: int, emb_channels: int, dropout: float, out_channels: Optional[int] = None, use_conv: bool = False,
                 use_scale_shift_norm: bool = False, dims: int = 2, use_checkpoint: bool = False, up: bool = False, down: bool = False,
                 kernel_size: int = 3, exchange_temb_dims: bool = False, skip_t_emb: bool = False):
        super().__init__()
        if use_scale_shift_norm or up or down or exchange_temb_dims or skip_t_emb or dropout != 0.0:
            raise NotImplementedError("ResBlock: scale-shift norm / resblock_updown / dropout>0 are not used by the SD/SDXL configs")
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels if out_channels is not None else channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        padding = kernel_size // 2
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding))
        self.updown = False
        self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(
            nn.GroupNorm(32, self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
            zero_module(conv_nd(dims, self.out_channels, self.out_channels, kernel_size, padding=padding)),
        )
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, kernel_size, padding=padding)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def _fwd(self, x: Img, emb: Tensor):
        gn1, gn2 = self.in_layers[0], self.out_layers[0]
        h1, b_gn1 = ops.groupnorm_fwd(x, gn1.weight, gn1.bias, 32, gn1.eps, silu=True)
        es, b_es = ops.silu_fwd(emb)
        eo, b_eo = linear_module_fwd(self.emb_layers[1], es)
        h2, b_c1 = self.in_layers[2].fwd(h1, rowvec=eo, stats_groups=32)      # the epilogue sums h2 for gn2: no statistics pass there
        h3, b_gn2 = ops.groupnorm_fwd(h2, gn2.weight, gn2.bias, 32, gn2.eps, silu=True)
        skip = self.skip_connection
        b_skip = None
        if isinstance(skip, nn.Identity):
            s = x.t
        elif skip.kernel_size == 1:
            s, b_skip = ops.linear_fwd(x.t, skip.weight, skip.bias)
        else:
            si, b_skip3 = skip.fwd(x)
            s, b_skip = si.t, (lambda g: b_skip3(g)[0].t)
        out, b_c2 = self.out_layers[3].fwd(h3, residual=s, stats_groups=32)   # ... and the block's output for the GroupNorm that reads it next

        def bwd(dy: Tensor):
            dh3, _ = b_c2(dy)
            dh2 = b_gn2(dh3.t)
            dh1, deo = b_c1(dh2)
            dxs = dy if b_skip is None else b_skip(dy)
            dx = b_gn1(dh1.t, dxs)
            demb = b_es(b_eo(deo))
            return dx, demb

        return out, bwd

    def fwd(self, x: Img, emb: Tensor):
        """bwd(dy tokens) -> (dx tokens, demb).  use_checkpoint drops and recomputes the block's activations
        (the reference wraps _forward in torch.utils.checkpoint, openaimodel.py:310-313)."""
        if not (self.use_checkpoint and ops.recording()):
            return self._fwd(x, emb)
        out, _ = self._fwd(x, emb)

        def bwd(dy: Tensor):
            _, b = self._fwd(x, emb)
            return b(dy)

        return out, bwd

    def forward(self, x: Tensor, emb: Tensor) -> Tensor:
        def run(x, emb):
            img = Img.from_nchw(x)
            out, bwd = self.fwd(img, as_tokens(emb))

            def bwd2(g):
                dx, demb = bwd(Img.from_nchw(g).t)
                return Img(dx, img.N, img.H, img.W).to_nchw(), demb.view(emb.shape)

            return out.to_nchw(), bwd2

        return apply_module(run, [x, emb], self)


class Timestep(nn.Module):
    """openaimodel.py (Timestep): sinusoidal embedding module."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, t: Tensor) -> Tensor:
        return ops.timestep_embedding(t, self.dim)


class UNetModel(nn.Module):
    """openaimodel.py:436-840.  Same constructor signature and module tree as the reference."""

    def __init__(
        self,
        in_channels: int,
        model_channels: int,
        out_channels: int,
        num_res_blocks: int,
        attention_resolutions,
        dropout: float = 0.0,
        channel_mult=(1, 2, 4, 8),
        conv_resample: bool = True,
        dims: int = 2,
        num_classes=None,
        use_checkpoint: bool = False,
        num_heads: int = -1,
        num_head_channels: int = -1,
        num_heads_upsample: int = -1,
        use_scale_shift_norm: bool = False,
        resblock_updown: bool = False,
        transformer_depth=1,
        context_dim: Optional[int] = None,
        disable_self_attentions: Optional[List[bool]] = None,
        num_attention_blocks: Optional[List[int]] = None,
        disable_middle_self_attn: bool = False,
        disable_middle_transformer: bool = False,
        use_linear_in_transformer: bool = False,
        spatial_transformer_attn_type: str = "softmax",
        adm_in_channels: Optional[int] = None,
    ):
        super().__init__()
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, "UNetModel: num_heads = -1 needs num_head_channels"
        if num_head_channels == -1:
            assert num_heads != -1, "UNetModel: num_head_channels = -1 needs num_heads"
        if resblock_updown:
            raise NotImplementedError("resblock_updown=True is not used by the SD/SDXL configs")
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(transformer_depth, int):
            transformer_depth = len(channel_mult) * [transformer_depth]
        transformer_depth_middle = transformer_depth[-1]
        if isinstance(num_res_blocks, int):
            self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        else:
            if len(num_res_blocks) != len(channel_mult):
                raise ValueError(f"UNetModel: num_res_blocks is an int or one entry per level ({len(channel_mult)}), got {num_res_blocks!r}")
            self.num_res_blocks = num_res_blocks
        if disable_self_attentions is not None and len(disable_self_attentions) != len(channel_mult):
            raise ValueError(f"UNetModel: disable_self_attentions needs one entry per level ({len(channel_mult)})")
        if num_attention_blocks is not None and len(num_attention_blocks) != len(self.num_res_blocks):
            raise ValueError("UNetModel: num_attention_blocks needs one entry per level, like num_res_blocks")

        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, time_embed_dim), nn.SiLU(), nn.Linear(time_embed_dim, time_embed_dim))

        if self.num_classes is not None:
            if self.num_classes == "sequential":
                if adm_in_channels is None:
                    raise ValueError("UNetModel: num_classes='sequential' needs adm_in_channels (the width of the vector conditioning)")
                self.label_emb = nn.Sequential(nn.Sequential(nn.Linear(adm_in_channels, time_embed_dim), nn.SiLU(), nn.Linear(time_embed_dim, time_embed_dim)))
            else:
                raise NotImplementedError(f"num_classes={self.num_classes!r}: only 'sequential' (SDXL) and None (SD1.5) are on the path")

        def make_st(ch, level_depth, disabled_sa):
            if num_head_channels == -1:
                heads, dim_head = num_heads, ch // num_heads
            else:
                heads, dim_head = ch // num_head_channels, num_head_channels
            return SpatialTransformer(ch, heads, dim_head, depth=level_depth, context_dim=context_dim, disable_self_attn=disabled_sa,
                                      use_linear=use_linear_in_transformer, attn_type=spatial_transformer_attn_type, use_checkpoint=use_checkpoint)

        def res(cin, cout):
            return ResBlock(cin, time_embed_dim, dropout, out_channels=cout, dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)

        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [res(ch, mult * model_channels)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if (context_dim is not None and disable_self_attentions is not None) else False
                    if num_attention_blocks is None or nr < num_attention_blocks[level]:
                        layers.append(make_st(ch, transformer_depth[level], disabled_sa))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                out_ch = ch
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=out_ch)))
                ch = out_ch
                input_block_chans.append(ch)
                ds *= 2

        self.middle_block = TimestepEmbedSequential(
            res(ch, ch),
            make_st(ch, transformer_depth_middle, disable_middle_self_attn) if not disable_middle_transformer else nn.Identity(),
            res(ch, ch),
        )

        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [res(ch + ich, model_channels * mult)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    disabled_sa = disable_self_attentions[level] if disable_self_attentions is not None else False
                    if num_attention_blocks is None or i < num_attention_blocks[level]:
                        layers.append(make_st(ch, transformer_depth[level], disabled_sa))
                if level and i == self.num_res_blocks[level]:
                    out_ch = ch
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=out_ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))

        self.out = nn.Sequential(nn.GroupNorm(32, ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        self.input_blocks[0]._nk_input_block = True
        # called as hook(module) right after each top-level block's parameter gradients have been ENQUEUED (backward order),
        # on the current stream and on ops.state.wgrad_stream: a consumer must order itself after both streams.  The
        # data-parallel wrapper uses it to start reducing that slice of the flat gradient buffer
        self.grad_ready_hook: Optional[Callable[[nn.Module], None]] = None
        self._nk_graphs = None   # neurosis_amd.graphs.ChainGraphs: hipGraph pairs of fwd / its backward, per input signature

    # -- the network as an explicit forward / backward chain over HIP kernels --------------------
    def _mlp_fwd(self, seq: nn.Sequential, x: Tensor, need_dx: bool):
        a, b0 = linear_module_fwd(seq[0], x, need_dx=need_dx)
        s, b1 = ops.silu_fwd(a)
        o, b2 = linear_module_fwd(seq[2], s)
        return o, (lambda g: b0(b1(b2(g))))

    def set_recompute(self, policy: Optional[str]) -> "UNetModel":
        """Selective activation recompute for every transformer block (BasicTransformerBlock.recompute): None keeps everything (default: 58 GB
        at SDXL batch 4 on a 288 GB part), "norms" rebuilds LayerNorm outputs and GEGLU products in backward instead of holding them.  The
        reference's own knob -- use_checkpoint, whole blocks through torch.utils.checkpoint -- is separate and still honoured."""
        from ..attention import BasicTransformerBlock

        if policy not in (None, "norms"):
            raise ValueError(f"recompute policy must be None or 'norms', got {policy!r}")
        for m in self.modules():
            if isinstance(m, BasicTransformerBlock):
                m.recompute = policy
        self._nk_graphs = None          # captured launch sequences belong to the old policy
        return self

    def _project_context(self, context: Tensor) -> None:
        """`to_k(context)`, `to_v(context)` of EVERY cross-attention of the network before the first block runs (reference
        modules/attention.py:383-385 computes them inside each block): the context does not depend on the UNet's activations, and
        one 308-row projection is 60 tiles on a 256-CU chip (24-26 us each, 70 per step).  Same-shape projections go out eight per
        launch (ops.gemm_nt_batched); each module picks its result up in CrossAttention.fwd."""
        from ..attention import BasicTransformerBlock
        from ...nn import adjacent

        if getattr(self, "_nk_cross", None) is None:
            self._nk_cross = [blk.attn2 for blk in self.modules() if isinstance(blk, BasicTransformerBlock)]
        groups = {}
        for att in self._nk_cross:
            wk, wv = att.to_k.weight, att.to_v.weight
            if wk.shape[1] != context.shape[1] or not adjacent(wk, wv):
                continue
            groups.setdefault((wk.shape[0], wk.shape[1]), []).append(att)
        for (inner, cdim), atts in groups.items():
            if len(atts) < 2:
                continue
            ws = [torch.as_strided(ops.shadow(a.to_k.weight), (2 * inner, cdim), (cdim, 1)) for a in atts]
            outs = ops.gemm_nt_batched([context] * len(atts), ws)
            for a, kv in zip(atts, outs):
                a._nk_kv = (kv, context)

    def fwd_graphed(self, x: Img, timesteps: Tensor, context: Optional[Tensor], y: Optional[Tensor]):
        """`fwd`, replayed from a hipGraph once this input signature has been seen twice (neurosis_amd/graphs.py): the training
        step's ~2 700 launches cost the host nothing.  NK_GRAPH=0 keeps the eager chain.  A gradient-ready hook (the
        data-parallel exchange) is called between the backward's per-block graph segments, as the eager chain calls it."""
        from ...graphs import ChainGraphs, frozen_stamp, graphs_enabled

        if not x.t.is_cuda or not graphs_enabled():
            return self.fwd(x, timesteps, context, y)
        if self._nk_graphs is None:
            self._nk_graphs = ChainGraphs(self.out[2].weight, hook=lambda: self.grad_ready_hook)
        N, H, W = x.N, x.H, x.W
        # (frozen_stamp: the captured kernels read the weights' bf16 shadows by address -- a flat store keeps them in place, free
        # parameters get new ones whenever they change; ~0.4 ms of host time per call for the SDXL UNet's 1 700 tensors)
        return self._nk_graphs.run(lambda t, ts, c, yy: self.fwd(Img(t, N, H, W), ts, c, yy), [x.t, timesteps, context, y],
                                   extra_key=(N, H, W, frozen_stamp(self), self.training))

    def fwd(self, x: Img, timesteps: Tensor, context: Optional[Tensor], y: Optional[Tensor]):
        """x: Img with channels padded to a multiple of 8.  Returns (out Img (padded channels), bwd);
        bwd(dout tokens) -> dx tokens or None."""
        hook_raw = self.grad_ready_hook
        t_emb = ops.timestep_embedding(timesteps, self.model_channels)
        emb, b_time = self._mlp_fwd(self.time_embed, t_emb, need_dx=False)
        b_label = None
        if self.num_classes is not None:
            lab, b_label = self._mlp_fwd(self.label_emb[0], y, need_dx=False)
            emb = ops.add(emb, lab)
        if context is not None:
            self._project_context(context)
        hs: List[Img] = []
        tape = []
        h = x
        for module in self.input_blocks:
            h, b = module.fwd(h, emb, context)
            hs.append(h)
            tape.append((module, b))
        h, b_mid = self.middle_block.fwd(h, emb, context)
        out_tape = []
        for module in self.output_blocks:
            h, b_cat = ops.cat_fwd(h, hs.pop())
            h, b = module.fwd(h, emb, context)
            out_tape.append((module, b, b_cat))
        gn = self.out[0]
        hn, b_gn = ops.groupnorm_fwd(h, gn.weight, gn.bias, 32, gn.eps, silu=True)
        out, b_conv = self.out[2].fwd(hn)

        def bwd(dout: Tensor):
            est = ops.state_of(self.out[2].weight)
            if est.segment_hook is not None:
                hook = est.segment_hook       # a hipGraph capture cuts the chain here; the replay calls the gradient-ready hook itself
            elif hook_raw is not None:
                def hook(m):
                    # the block's weight gradients may still be in flight on the side stream: the hook's owner waits for that
                    # stream itself (FlatDataParallel does, on its exchange stream), so backward is not stalled here
                    hook_raw(m)
            else:
                hook = None
            dh, _ = b_conv(dout)
            dh = b_gn(dh.t)
            if hook:
                hook(self.out)
            demb = None

            def acc_emb(de):
                nonlocal demb
                if de is not None:
                    demb = de if demb is None else ops.add(demb, de)

            skips = []
            for module, b, b_cat in reversed(out_tape):
                dh, de = b(dh)
                acc_emb(de)
                dh, dskip = b_cat(dh)
                skips.append(dskip)
                if hook:
                    hook(module)
            out_tape.clear()
            dh, de = b_mid(dh)
            acc_emb(de)
            if hook:
                hook(self.middle_block)
            # skips[i] is the gradient that reached input block i's output through its skip connection
            for module, b in reversed(tape):
                dh = ops.add(dh, skips.pop())
                dh, de = b(dh)
                acc_emb(de)
                if hook:
                    hook(module)
            tape.clear()
            if b_label is not None:
                b_label(demb)
                if hook:
                    hook(self.label_emb)
            b_time(demb)
            if hook:
                hook(self.time_embed)
            ops.join_wgrad_stream(self.out[2].weight)
            return dh

        return out, bwd

    def forward(self, x: Tensor, timesteps: Optional[Tensor] = None, context: Optional[Tensor] = None, y: Optional[Tensor] = None, **kwargs) -> Tensor:
        """openaimodel.py:803-840: x [N, C, H, W], timesteps [N], context [N, L, context_dim], y [N, adm_in_channels]."""
        if (y is not None) != (self.num_classes is not None):
            raise ValueError(f"UNetModel.forward: this network has no label embedding (num_classes=None) but was given y of shape {tuple(y.shape)}")
        if y is not None:
            assert y.shape[0] == x.shape[0]
        N, Cin, H, W = x.shape
        ins = [x, timesteps, context, y]

        def run(x, timesteps, context, y):
            cpad = (Cin + 7) // 8 * 8
            if cpad == Cin:
                img = Img.from_nchw(x)
            else:
                img = Img(ops.nchw_to_tokens(x, cpad), N, H, W)
            out, bwd = self.fwd(img, timesteps, None if context is None else as_tokens(context), None if y is None else as_tokens(y))
            o = ops.tokens_to_nchw(out.t, N, self.out_channels, out.H, out.W, dtype=x.dtype if x.dtype in (torch.float32, BF16) else torch.float32)

            def bwd2(g):
                gt = ops.nchw_to_tokens(g, out.C)
                dx = bwd(gt)
                gx = None
                if dx is not None and x.requires_grad:
                    gx = ops.tokens_to_nchw(dx, N, Cin, H, W, dtype=x.dtype)
                return gx, None, None, None

            return o, bwd2

        return apply_module(run, ins, self)
