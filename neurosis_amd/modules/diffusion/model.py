"""VAE Encoder and Decoder on MI355X: the class surface of neurosis.modules.diffusion.model.{Encoder,Decoder}
(/root/reference/src/neurosis/modules/diffusion/model.py:456-606 and :609-765) over the HIP kernels.

Same constructor arguments, module tree and state_dict keys (conv_in, down.{l}.block.{i}.{norm1,conv1,norm2,conv2,
nin_shortcut}, down.{l}.downsample.conv, up.{l}.block.{i}.*, up.{l}.upsample.conv, mid.{block_1,attn_1,block_2}, norm_out,
conv_out, quant_conv / post_quant_conv).  Two ways in: `fwd` / `forward` -- forward only, what the diffusion engine runs
under no_grad (DiffusionEngine.encode_first_stage / decode_first_stage, models/diffusion.py:172-197) -- and `fwdb`, which also
returns the backward closure of the block (weight gradients into the parameters' .grad, input gradient returned) for
training the autoencoder itself (models/autoencoder.AutoencodingEngine; SURVEY 8(f) N2, reconstruction part).
"""
from __future__ import annotations

import os
import math
from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from ... import ops
from ...graphs import ForwardGraphs, frozen_stamp, graphs_enabled
from ...nn import Conv2d
from ...ops import BF16, Img


def Normalize(in_channels: int, num_groups: int = 32) -> nn.GroupNorm:
    """modules/layers.py:5-7."""
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _gn(x: Img, norm: nn.GroupNorm, silu: bool) -> Img:
    return ops.groupnorm_fwd(x, norm.weight, norm.bias, norm.num_groups, norm.eps, silu)[0]


def _conv_fwdb(conv: Conv2d, x: Img, **kw):
    """(y Img, bwd) with bwd(dy tokens) -> dx tokens"""
    y, b = conv.fwd(x, **kw)
    return y, lambda dy: b(dy)[0].t


class Downsample(nn.Module):
    """model.py:65-82: ConstantPad2d((0,1,0,1)) + 3x3 stride-2 conv, as one implicit-GEMM gather."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by the SD/SDXL VAE")
        self.with_conv = with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0, asym_pad=True)

    def fwd(self, x: Img) -> Img:
        return self.conv.fwd(x, need_dx=False)[0]

    def fwdb(self, x: Img):
        return _conv_fwdb(self.conv, x)


class Upsample(nn.Module):
    """model.py:44-62: nearest-neighbour x2 followed by a 3x3 conv.  The conv's implicit-GEMM gather reads the
    low-resolution tensor at (y >> 1, x >> 1), so the upsampled image never exists in HBM."""

    def __init__(self, in_channels: int, with_conv: bool):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("resamp_with_conv=False is not used by the SD/SDXL VAE")
        self.with_conv = with_conv
        self.conv = Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def fwd(self, x: Img) -> Img:
        return self.conv.fwd(x, upsample=True, need_dx=False)[0]

    def fwdb(self, x: Img):
        return _conv_fwdb(self.conv, x, upsample=True)


class ResnetBlock(nn.Module):
    """model.py:85-134 with temb=None."""

    def __init__(self, *, in_channels: int, out_channels: Optional[int] = None, conv_shortcut: bool = False, dropout: float = 0.0, temb_channels: int = 512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Identity()
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def fwd(self, x: Img, want_sums: bool = False) -> Img:
        """forward only (the frozen first stage).  Each convolution's epilogue emits the GroupNorm sums of its output (conv1 for norm2;
        with `want_sums` conv2 for whichever GroupNorm reads the block's output), so the GroupNorms run their normalisation pass only.
        (Applying the GroupNorm inside the consuming convolution, as an in-LDS rewrite of its halo, was built and measured slower than
        the separate normalisation pass -- DESIGN.md -- and removed.)"""
        groups = self.norm1.num_groups
        h = self.conv1.fwd(_gn(x, self.norm1, True), need_dx=False, stats_groups=self.norm2.num_groups)[0]
        h = _gn(h, self.norm2, True)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                s = self.conv_shortcut.fwd(x, need_dx=False)[0].t
            else:
                s = ops.gemm_nt(x.t, ops.w2d(self.nin_shortcut.weight), self.nin_shortcut.bias)
        else:
            s = x.t
        return self.conv2.fwd(h, residual=s, need_dx=False, stats_groups=groups if want_sums else None)[0]

    def fwdb(self, x: Img):
        """(y, bwd); bwd(dy tokens) -> dx tokens"""
        n1, n2 = self.norm1, self.norm2
        h0, b_n1 = ops.groupnorm_fwd(x, n1.weight, n1.bias, n1.num_groups, n1.eps, True)
        h1, b_c1 = self.conv1.fwd(h0)
        h2, b_n2 = ops.groupnorm_fwd(h1, n2.weight, n2.bias, n2.num_groups, n2.eps, True)
        b_short = None
        if self.in_channels != self.out_channels:
            short, b_short = (self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut).fwd(x)
            skip = short.t
        else:
            skip = x.t
        y, b_c2 = self.conv2.fwd(h2, residual=skip)

        def bwd(dy: Tensor) -> Tensor:
            dh1 = b_n2(b_c2(dy)[0].t)
            dh0 = b_c1(dh1)[0].t
            through_skip = dy if b_short is None else b_short(dy)[0].t
            return b_n1(dh0, dx_add=through_skip)

        return y, bwd


class AttnBlock(nn.Module):
    """model.py:144-243 (AttnBlock / MemoryEfficientAttnBlock / TorchSDPAttnBlock): single-head self-attention over
    H*W tokens with head dim = C.  C = 512 (the SD / SDXL VAE) runs the one-kernel flash forward of csrc/attn512.h -- no [L][L] score
    matrix in HBM --, C <= 160 the flash kernels of the UNet (forward and backward); other widths (inference only) go through two MFMA
    GEMMs around a row softmax.  The backward at C = 512 (autoencoder training) recomputes the probabilities per chunk of query rows
    (ops.attention512_fwd): nothing of size L x L is kept between forward and backward."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def fwd(self, x: Img) -> Img:
        hn = _gn(x, self.norm, False)
        q, k, v = (ops.gemm_nt(hn.t, ops.w2d(m.weight), m.bias) for m in (self.q, self.k, self.v))
        C = self.in_channels
        fused = (C == 512 and os.environ.get("NK_ATTN512", "1") != "0") or C <= 160      # NK_ATTN512=0: A/B switch
        o = ops.attention_fwd(q, k, v, x.N, 1, C, need_lse=False)[0] if fused else ops.attention_unfused(q, k, v, x.N)
        y = ops.gemm_nt(o, ops.w2d(self.proj_out.weight), self.proj_out.bias, residual=x.t)
        return Img(y, x.N, x.H, x.W)

    def fwdb(self, x: Img):
        n = self.norm
        hn, b_n = ops.groupnorm_fwd(x, n.weight, n.bias, n.num_groups, n.eps, False)
        (q, b_q), (k, b_k), (v, b_v) = (m.fwd(hn) for m in (self.q, self.k, self.v))
        if self.in_channels <= 160:
            o, b_att = ops.attention_fwd(q.t, k.t, v.t, x.N, 1, self.in_channels)
        elif self.in_channels == 512:
            o, b_att = ops.attention512_fwd(q.t, k.t, v.t, x.N)       # flash forward; backward recomputes the probabilities chunk by chunk
        else:       # any other width (a VAE whose last level has 256 or 384 channels): two-GEMM forward, chunked recompute backward
            o, b_att = ops.attention_anydim_fwd(q.t, k.t, v.t, x.N)
        y, b_p = self.proj_out.fwd(Img(o, x.N, x.H, x.W), residual=x.t)

        def bwd(dy: Tensor) -> Tensor:
            dq, dk, dv = b_att(b_p(dy)[0].t)
            dhn = ops.add(ops.add(b_q(dq)[0].t, b_k(dk)[0].t), b_v(dv)[0].t)
            return b_n(dhn, dx_add=dy)

        return y, bwd


MemoryEfficientAttnBlock = AttnBlock


def make_attn(in_channels: int, attn_type: str = "vanilla", attn_kwargs=None) -> nn.Module:
    """model.py:255-283.  "vanilla" (AttnBlock) and "vanilla-xformers" (MemoryEfficientAttnBlock, what the SD/SDXL
    configs name) are the same function and map to the HIP AttnBlock.  The reference's "torch-sdp" block
    (TorchSDPAttnBlock, model.py:224-243) views the NCHW q/k/v memory as (B, HW, 1, C) WITHOUT permuting, i.e. it
    attends over scrambled tokens -- a different function that no shipped config selects; it is refused rather than
    silently replaced (SURVEY quirk list)."""
    if attn_type in ("vanilla", "vanilla-xformers"):
        return AttnBlock(in_channels)
    if attn_type == "none":
        return nn.Identity()
    if attn_type == "torch-sdp":
        raise ValueError("attn_type 'torch-sdp' (TorchSDPAttnBlock) scrambles tokens in the reference; use 'vanilla' or 'vanilla-xformers'")
    raise ValueError(f"attn_type {attn_type} unknown or outside the SD/SDXL path")


class Encoder(nn.Module):
    """model.py:456-606."""

    def __init__(self, *, ch: int, out_ch: int, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int, attn_resolutions: Sequence[int],
                 dropout: float = 0.0, resamp_with_conv: bool = True, in_channels: int, resolution: int, z_channels: int, double_z: bool = True,
                 use_linear_attn: bool = False, attn_type: str = "vanilla", embed_dim: int = 256, standalone: bool = False, **kwargs):
        super().__init__()
        self.ch = ch
        self.temb_ch = 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.z_channels = z_channels
        self.double_z = double_z
        self.conv_in = Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            attn = nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block = block
            down.attn = attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)
        self.max_batch_size = None
        self.standalone = standalone
        if self.standalone:
            qc = (1 + double_z) * z_channels
            self.quant_conv = Conv2d(qc, (1 + double_z) * embed_dim, 1)
        else:
            self.quant_conv = nn.Identity()

    def fwd(self, x: Img) -> Img:
        """encode + quant_conv on an Img whose channels are already padded to a multiple of 8."""
        h = self.conv_in.fwd(x, need_dx=False)[0]
        last_level = self.num_resolutions - 1
        for i_level in range(self.num_resolutions):
            level = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                # the block's output feeds a GroupNorm (the next block's norm1, an attention block's norm, mid.block_1.norm1) unless a
                # Downsample comes next: ask its last convolution for that GroupNorm's sums
                to_norm = i_block + 1 < self.num_res_blocks or len(level.attn) > 0 or i_level == last_level
                h = level.block[i_block].fwd(h, want_sums=to_norm)
                if len(level.attn) > 0:
                    h = level.attn[i_block].fwd(h)
            if i_level != last_level:
                h = level.downsample.fwd(h)
        h = self.mid.block_1.fwd(h, want_sums=True)
        if not isinstance(self.mid.attn_1, nn.Identity):
            h = self.mid.attn_1.fwd(h)
        h = self.mid.block_2.fwd(h, want_sums=True)
        h = self.conv_out.fwd(_gn(h, self.norm_out, True), need_dx=False)[0]
        if self.standalone:
            h = self.quant_conv.fwd(h, need_dx=False)[0]
        return h

    def fwdb(self, x: Img):
        """encode (+ quant_conv when standalone) keeping the backward: (moments Img, bwd); bwd(d_moments tokens) -> None (the
        image needs no gradient)."""
        tape = []

        def run(pair):
            tape.append(pair[1])
            return pair[0]

        h, b_in = self.conv_in.fwd(x, need_dx=False)
        for i_level in range(self.num_resolutions):
            level = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                h = run(level.block[i_block].fwdb(h))
                if len(level.attn) > 0:
                    h = run(level.attn[i_block].fwdb(h))
            if i_level != self.num_resolutions - 1:
                h = run(level.downsample.fwdb(h))
        h = run(self.mid.block_1.fwdb(h))
        if not isinstance(self.mid.attn_1, nn.Identity):
            h = run(self.mid.attn_1.fwdb(h))
        h = run(self.mid.block_2.fwdb(h))
        no = self.norm_out
        hn, b_no = ops.groupnorm_fwd(h, no.weight, no.bias, no.num_groups, no.eps, True)
        h, b_out = self.conv_out.fwd(hn)
        b_quant = None
        if self.standalone:
            h, b_quant = self.quant_conv.fwd(h)

        def bwd(dh: Tensor) -> None:
            if b_quant is not None:
                dh = b_quant(dh)[0].t
            dh = b_no(b_out(dh)[0].t)
            for b in reversed(tape):
                dh = b(dh)
            tape.clear()
            b_in(dh)

        return h, bwd

    @torch.no_grad()
    def forward(self, x: Tensor, regularize: bool = False) -> Tensor:
        """model.py:585-606.  With regularize=True returns the DiagonalGaussian mode = the mean half of the
        moments (regularizers.py:31-41, distributions.py:28-37,71-72; the discarded KL term is not computed).
        Returns fp32 NCHW."""
        N, Cin, H, W = x.shape
        bs = self.max_batch_size or N
        outs = []
        for i in range(0, N, bs):
            xb = x[i:i + bs]
            n = xb.shape[0]
            xb = xb.float() if xb.dtype not in (torch.float32, BF16) else xb
            if xb.is_cuda and graphs_enabled("vae") and not any(p.requires_grad for p in self.parameters()):
                # the frozen encoder of the training step: ~600 launches replayed from a hipGraph (neurosis_amd/graphs.py)
                fg = self.__dict__.get("_nk_fgraphs")
                if fg is None:
                    fg = self.__dict__["_nk_fgraphs"] = ForwardGraphs(xb.device)
                outs.append(fg.run(lambda t: self._encode_chunk(t, regularize), [xb.contiguous()], extra_key=(regularize, frozen_stamp(self))))
            else:
                outs.append(self._encode_chunk(xb, regularize))
        return outs[0] if len(outs) == 1 else torch.cat(outs, 0)

    def _encode_chunk(self, xb: Tensor, regularize: bool) -> Tensor:
        n, Cin, H, W = xb.shape
        img = Img(ops.nchw_to_tokens(xb, (Cin + 7) // 8 * 8), n, H, W)
        h = self.fwd(img)
        zc_real = self.conv_out.out_channels if not self.standalone else self.quant_conv.out_channels
        keep = zc_real // 2 if (regularize and self.double_z) else zc_real
        return ops.tokens_to_nchw(h.t, n, keep, h.H, h.W, dtype=torch.float32)


class Decoder(nn.Module):
    """model.py:609-765.  Latents in, image out; levels are built coarse to fine but stored fine-first (`up[0]` is the
    full-resolution level), which is what the checkpoint keys expect."""

    def __init__(self, *, ch: int, out_ch: int, ch_mult: Sequence[int] = (1, 2, 4, 8), num_res_blocks: int, attn_resolutions: Sequence[int],
                 dropout: float = 0.0, resamp_with_conv: bool = True, in_channels: int, resolution: int, z_channels: int, give_pre_end: bool = False,
                 tanh_out: bool = False, use_linear_attn: bool = False, attn_type: str = "vanilla", embed_dim: int = 256, standalone: bool = False,
                 **kwargs):
        super().__init__()
        if use_linear_attn:
            raise ValueError("linear attention is outside the SD/SDXL path")
        levels = len(ch_mult)
        self.ch, self.temb_ch, self.num_resolutions, self.num_res_blocks = ch, 0, levels, num_res_blocks
        self.resolution, self.in_channels, self.out_ch = resolution, in_channels, out_ch
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        width = ch * ch_mult[-1]
        res = resolution // 2 ** (levels - 1)
        self.z_shape = (1, z_channels, res, res)
        self.conv_in = Conv2d(z_channels, width, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(width, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=width, out_channels=width, temb_channels=0, dropout=dropout)
        stages = []
        for level in range(levels - 1, -1, -1):
            stage = nn.Module()
            stage.block, stage.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                stage.block.append(ResnetBlock(in_channels=width, out_channels=ch * ch_mult[level], temb_channels=0, dropout=dropout))
                width = ch * ch_mult[level]
                if res in attn_resolutions:
                    stage.attn.append(make_attn(width, attn_type=attn_type))
            if level > 0:
                stage.upsample = Upsample(width, resamp_with_conv)
                res *= 2
            stages.append(stage)
        self.up = nn.ModuleList(reversed(stages))
        self.norm_out = Normalize(width)
        self.conv_out = Conv2d(width, out_ch, kernel_size=3, stride=1, padding=1)
        self.max_batch_size = None
        self.standalone = standalone
        self.post_quant_conv = Conv2d(embed_dim, z_channels, 1) if standalone else nn.Identity()

    def get_last_layer(self, **kwargs):
        return self.conv_out.weight

    def fwd(self, z: Img) -> Img:
        """post_quant_conv (when standalone) + decode on an Img whose channels are padded to a multiple of 8"""
        if self.standalone:
            z = self.post_quant_conv.fwd(z, need_dx=False)[0]
        h = self.conv_in.fwd(z, need_dx=False)[0]
        h = self.mid.block_1.fwd(h)
        if not isinstance(self.mid.attn_1, nn.Identity):
            h = self.mid.attn_1.fwd(h)
        h = self.mid.block_2.fwd(h)
        for level in range(self.num_resolutions - 1, -1, -1):
            stage = self.up[level]
            for i, block in enumerate(stage.block):
                h = block.fwd(h)
                if len(stage.attn) > 0:
                    h = stage.attn[i].fwd(h)
            if level > 0:
                h = stage.upsample.fwd(h)
        if self.give_pre_end:
            return h
        return self.conv_out.fwd(_gn(h, self.norm_out, True), need_dx=False)[0]

    def fwdb(self, z: Img):
        """(image Img, bwd); bwd(d_image tokens) -> dz tokens"""
        if self.give_pre_end or self.tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not used by the SD/SDXL autoencoder configs")
        tape = []

        def run(pair):
            tape.append(pair[1])
            return pair[0]

        h = z
        if self.standalone:
            h = run(_conv_fwdb(self.post_quant_conv, h))
        h = run(_conv_fwdb(self.conv_in, h))
        h = run(self.mid.block_1.fwdb(h))
        if not isinstance(self.mid.attn_1, nn.Identity):
            h = run(self.mid.attn_1.fwdb(h))
        h = run(self.mid.block_2.fwdb(h))
        for level in range(self.num_resolutions - 1, -1, -1):
            stage = self.up[level]
            for i, block in enumerate(stage.block):
                h = run(block.fwdb(h))
                if len(stage.attn) > 0:
                    h = run(stage.attn[i].fwdb(h))
            if level > 0:
                h = run(stage.upsample.fwdb(h))
        no = self.norm_out
        hn, b_no = ops.groupnorm_fwd(h, no.weight, no.bias, no.num_groups, no.eps, True)
        out, b_out = self.conv_out.fwd(hn)

        def bwd(dout: Tensor) -> Tensor:
            dh = b_no(b_out(dout)[0].t)
            for b in reversed(tape):
                dh = b(dh)
            tape.clear()
            return dh

        def last_layer_grad_norm(dout: Tensor) -> Tensor:
            """|| d<dout, image> / d conv_out.weight ||  (device scalar): the quantity the adversarial loss balances its two
            terms with (GeneralLPIPSWithDiscriminator.calculate_adaptive_weight, discriminator_loss.py:205-217).  Runs only the
            last convolution's weight-gradient kernel; call before `bwd`, which overwrites the gradient."""
            b_out(dout)
            ops.join_wgrad_stream()
            return self.conv_out.weight.grad.float().norm()

        bwd.last_layer_grad_norm = last_layer_grad_norm
        return out, bwd

    @torch.no_grad()
    def forward(self, z: Tensor, cat_zero: bool = False, **kwargs):
        """model.py:746-765.  fp32 NCHW out.  With `max_batch_size` set the reference returns a python list of chunk outputs
        unless cat_zero=True; that is kept."""
        N, Cz, H, W = z.shape
        bs = self.max_batch_size or N
        outs = []
        for i in range(0, N, bs):
            zb = z[i:i + bs]
            n = zb.shape[0]
            img = Img(ops.nchw_to_tokens(zb.float() if zb.dtype not in (torch.float32, BF16) else zb, (Cz + 7) // 8 * 8), n, H, W)
            h = self.fwd(img)
            keep = h.C if self.give_pre_end else self.out_ch
            o = ops.tokens_to_nchw(h.t, n, keep, h.H, h.W, dtype=torch.float32)
            outs.append(torch.tanh(o) if self.tanh_out and not self.give_pre_end else o)
        if self.max_batch_size is None:
            return outs[0]
        return torch.cat(outs, 0) if cat_zero else outs
