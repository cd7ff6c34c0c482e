"""CPU oracle for the SDXL training-step hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
path (neurosis_amd/) never does and fails loudly if its HIP library is missing.

It is a functional, fp32, plain-PyTorch-CPU restatement of the reference's algorithm for the path
(floating-point arithmetic, so the oracle is a torch fp32 reference as the task allows), written against a
`state_dict` with the reference's own parameter names.  Every function cites the reference lines it follows
(paths relative to /root/reference/src/neurosis/).  Gradients come from torch autograd on the CPU.

Parity status: the reference has no tests, golden vectors or fixtures of its own, so
this oracle is pinned by golden vectors captured from the reference itself, imported in the authoring
container by tests/golden/make_golden.py (committed together with the fixtures) -- see
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

SD = dict  # state_dict: name -> fp32 tensor


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def timestep_embedding(timesteps: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    """modules/diffusion/util.py:152-177 (repeat_only=False): [cos | sin], zero pad if dim is odd."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def linear(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def conv(sd: SD, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def group_norm(sd: SD, p: str, x: Tensor, eps: float) -> Tensor:
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def layer_norm(sd: SD, p: str, x: Tensor) -> Tensor:
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], 1e-5)


# ------------------------------------------------------------------------------------------------
# UNet blocks
# ------------------------------------------------------------------------------------------------
def resblock(sd: SD, p: str, x: Tensor, emb: Tensor) -> Tensor:
    """ResBlock._forward, modules/diffusion/openaimodel.py:315-342 (no up/down, no scale-shift norm; GN eps 1e-5)."""
    h = conv(sd, p + ".in_layers.2", F.silu(group_norm(sd, p + ".in_layers.0", x, 1e-5)))
    emb_out = linear(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + emb_out[:, :, None, None]
    h = conv(sd, p + ".out_layers.3", F.silu(group_norm(sd, p + ".out_layers.0", h, 1e-5)))
    if (p + ".skip_connection.weight") in sd:
        x = conv(sd, p + ".skip_connection", x, padding=0)
    return x + h


def attention(sd: SD, p: str, x: Tensor, context: Optional[Tensor], heads: int) -> Tensor:
    """TorchSDPCrossAttention.forward, modules/attention.py:369-417: q/k/v without bias, softmax(qk^T/sqrt(d))v, to_out.0."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, _, inner = q.shape
    d = inner // heads
    q, k, v = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)  # attention.py:410-412
    o = o.transpose(1, 2).reshape(b, -1, inner)
    return linear(sd, p + ".to_out.0", o)


def feed_forward(sd: SD, p: str, x: Tensor) -> Tensor:
    """FeedForward with GEGLU, modules/attention.py:50-74: proj -> chunk -> a*gelu_erf(gate) -> Linear."""
    a, gate = linear(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(sd, p + ".net.2", a * F.gelu(gate))


def transformer_block(sd: SD, p: str, x: Tensor, context: Optional[Tensor], heads: int) -> Tensor:
    """BasicTransformerBlock._forward, modules/attention.py:487-511 (disable_self_attn=False)."""
    x = x + attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None, heads)
    x = x + attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), context, heads)
    x = x + feed_forward(sd, p + ".ff", layer_norm(sd, p + ".norm3", x))
    return x


def spatial_transformer(sd: SD, p: str, x: Tensor, context: Optional[Tensor], heads: int, depth: int, use_linear: bool) -> Tensor:
    """SpatialTransformer.forward, modules/attention.py:642-667 (GroupNorm eps 1e-6)."""
    b, c, h, w = x.shape
    x_in = x
    x = group_norm(sd, p + ".norm", x, 1e-6)
    if not use_linear:
        x = conv(sd, p + ".proj_in", x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
    if use_linear:
        x = linear(sd, p + ".proj_in", x)
    for d in range(depth):
        x = transformer_block(sd, f"{p}.transformer_blocks.{d}", x, context, heads)
    if use_linear:
        x = linear(sd, p + ".proj_out", x)
    x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
    if not use_linear:
        x = conv(sd, p + ".proj_out", x, padding=0)
    return x + x_in


def unet_plan(cfg: dict):
    """Topology of UNetModel.__init__, modules/diffusion/openaimodel.py:620-795, as a list of
    (block name, [layer kinds]) -- the same loop structure, without building modules."""
    mc = cfg["model_channels"]
    mult = list(cfg.get("channel_mult", (1, 2, 4, 8)))
    nrb = cfg["num_res_blocks"]
    nrb = [nrb] * len(mult) if isinstance(nrb, int) else list(nrb)
    att = list(cfg["attention_resolutions"])
    td = cfg.get("transformer_depth", 1)
    td = [td] * len(mult) if isinstance(td, int) else list(td)
    nh, nhc = cfg.get("num_heads", -1), cfg.get("num_head_channels", -1)

    def heads_of(ch):
        return ch // nhc if nhc != -1 else nh

    inp = [("input_blocks.0", [("conv", None)])]
    chans = [mc]
    ch, ds, idx = mc, 1, 1
    for level, m in enumerate(mult):
        for _ in range(nrb[level]):
            layers = [("res", None)]
            ch = m * mc
            if ds in att:
                layers.append(("st", (heads_of(ch), td[level])))
            inp.append((f"input_blocks.{idx}", layers))
            idx += 1
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append((f"input_blocks.{idx}", [("down", None)]))
            idx += 1
            chans.append(ch)
            ds *= 2
    mid = [("res", None), ("st", (heads_of(ch), td[-1])), ("res", None)]
    out = []
    idx = 0
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb[level] + 1):
            chans.pop()
            layers = [("res", None)]
            ch = mc * m
            if ds in att:
                layers.append(("st", (heads_of(ch), td[level])))
            if level and i == nrb[level]:
                layers.append(("up", None))
                ds //= 2
            out.append((f"output_blocks.{idx}", layers))
            idx += 1
    return inp, mid, out


def _run_layers(sd: SD, name: str, layers, h: Tensor, emb: Tensor, context, use_linear: bool) -> Tensor:
    """TimestepEmbedSequential.forward dispatch, modules/diffusion/openaimodel.py:71-93."""
    for j, (kind, arg) in enumerate(layers):
        p = f"{name}.{j}"
        if kind == "conv":
            h = conv(sd, p, h)
        elif kind == "res":
            h = resblock(sd, p, h, emb)
        elif kind == "st":
            h = spatial_transformer(sd, p, h, context, arg[0], arg[1], use_linear)
        elif kind == "down":  # Downsample.forward :195-197, 3x3 stride-2 pad-1 conv
            h = conv(sd, p + ".op", h, stride=2)
        elif kind == "up":  # Upsample.forward :126-143, nearest x2 then 3x3 conv
            h = conv(sd, p + ".conv", F.interpolate(h, scale_factor=2, mode="nearest"))
    return h


def unet_forward(sd: SD, cfg: dict, x: Tensor, timesteps: Tensor, context: Optional[Tensor], y: Optional[Tensor]) -> Tensor:
    """UNetModel.forward, modules/diffusion/openaimodel.py:803-840."""
    use_linear = cfg.get("use_linear_in_transformer", False)
    inp, mid, out = unet_plan(cfg)
    t_emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = linear(sd, "time_embed.2", F.silu(linear(sd, "time_embed.0", t_emb)))
    if cfg.get("num_classes") is not None:
        assert cfg["num_classes"] == "sequential", "oracle covers the SDXL 'sequential' label embedding"
        emb = emb + linear(sd, "label_emb.0.2", F.silu(linear(sd, "label_emb.0.0", y)))
    hs = []
    h = x
    for name, layers in inp:
        h = _run_layers(sd, name, layers, h, emb, context, use_linear)
        hs.append(h)
    h = _run_layers(sd, "middle_block", mid, h, emb, context, use_linear)
    for name, layers in out:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, name, layers, h, emb, context, use_linear)
    h = F.silu(group_norm(sd, "out.0", h, 1e-5))
    return conv(sd, "out.2", h)


# ------------------------------------------------------------------------------------------------
# VAE encoder
# ------------------------------------------------------------------------------------------------
def vae_resnet(sd: SD, p: str, x: Tensor) -> Tensor:
    """ResnetBlock.forward with temb=None, modules/diffusion/model.py:114-134 (Normalize = GN32 eps 1e-6, layers.py:5-7)."""
    h = conv(sd, p + ".conv1", F.silu(group_norm(sd, p + ".norm1", x, 1e-6)))
    h = conv(sd, p + ".conv2", F.silu(group_norm(sd, p + ".norm2", h, 1e-6)))
    if (p + ".nin_shortcut.weight") in sd:
        x = conv(sd, p + ".nin_shortcut", x, padding=0)
    return x + h


def vae_attn(sd: SD, p: str, x: Tensor) -> Tensor:
    """AttnBlock / TorchSDPAttnBlock, modules/diffusion/model.py:144-172,224-243: single head over H*W tokens, d = C."""
    b, c, h, w = x.shape
    hn = group_norm(sd, p + ".norm", x, 1e-6)
    q, k, v = (conv(sd, f"{p}.{n}", hn, padding=0).permute(0, 2, 3, 1).reshape(b, h * w, c) for n in ("q", "k", "v"))
    s = torch.matmul(q, k.transpose(1, 2)) * (c ** -0.5)
    o = torch.matmul(torch.softmax(s, dim=-1), v).reshape(b, h, w, c).permute(0, 3, 1, 2)
    return x + conv(sd, p + ".proj_out", o, padding=0)


def vae_encode(sd: SD, dd: dict, x: Tensor) -> Tensor:
    """Encoder.forward(x, regularize=True), modules/diffusion/model.py:558-606 with quant_conv (:551-556,592) and
    DiagonalGaussianRegularizer(sample=False) -> distribution mode = first half of the channels
    (regularizers.py:31-41, distributions.py:28-37,71-72).  Returns the mean (B, z_channels, H/8, W/8)."""
    ch_mult = list(dd["ch_mult"])
    nrb = dd["num_res_blocks"]
    h = conv(sd, "conv_in", x)
    for lvl in range(len(ch_mult)):
        for ib in range(nrb):
            h = vae_resnet(sd, f"down.{lvl}.block.{ib}", h)
            if f"down.{lvl}.attn.{ib}.norm.weight" in sd:
                h = vae_attn(sd, f"down.{lvl}.attn.{ib}", h)
        if lvl != len(ch_mult) - 1:  # Downsample.forward :76-82: pad (0,1,0,1) then 3x3 stride 2 pad 0
            h = conv(sd, f"down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = vae_resnet(sd, "mid.block_1", h)
    h = vae_attn(sd, "mid.attn_1", h)
    h = vae_resnet(sd, "mid.block_2", h)
    h = conv(sd, "conv_out", F.silu(group_norm(sd, "norm_out", h, 1e-6)))
    if "quant_conv.weight" in sd:
        h = conv(sd, "quant_conv", h, padding=0)
    mean, _logvar = torch.chunk(h, 2, dim=1)
    return mean


def vae_moments(sd: SD, dd: dict, x: Tensor) -> Tensor:
    """Encoder.forward(x, regularize=False) incl. quant_conv: the [mean | logvar] moments (model.py:558-606)."""
    ch_mult = list(dd["ch_mult"])
    h = conv(sd, "conv_in", x)
    for lvl in range(len(ch_mult)):
        for ib in range(dd["num_res_blocks"]):
            h = vae_resnet(sd, f"down.{lvl}.block.{ib}", h)
            if f"down.{lvl}.attn.{ib}.norm.weight" in sd:
                h = vae_attn(sd, f"down.{lvl}.attn.{ib}", h)
        if lvl != len(ch_mult) - 1:
            h = conv(sd, f"down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = vae_resnet(sd, "mid.block_2", vae_attn(sd, "mid.attn_1", vae_resnet(sd, "mid.block_1", h)))
    h = conv(sd, "conv_out", F.silu(group_norm(sd, "norm_out", h, 1e-6)))
    return conv(sd, "quant_conv", h, padding=0) if "quant_conv.weight" in sd else h


def vae_reconstruction_loss(enc_sd: SD, dec_sd: SD, dd: dict, x: Tensor, noise: Tensor, kl_weight: float = 0.0):
    """AutoencodingEngine.forward (models/autoencoder.py:222-225) with DiagonalGaussianRegularizer(sample=True)
    (modules/regularizers.py:23-42, modules/distributions.py:28-60; the sample's noise injected) and the engine's simple-loss
    branch (:247-256) with nn.MSELoss; `kl_weight` adds regularization_weights["kl_loss"] * kl_loss as
    GeneralLPIPSWithDiscriminator does (discriminator_loss.py:283-286).  Returns (loss, moments, z, xrec, kl_loss)."""
    moments = vae_moments(enc_sd, dd, x)
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z = mean + torch.exp(0.5 * logvar) * noise
    kl = 0.5 * torch.sum(mean ** 2 + torch.exp(logvar) - 1.0 - logvar, dim=[1, 2, 3])
    kl_loss = kl.sum() / kl.shape[0]
    xrec = vae_decode(dec_sd, dd, z)
    return F.mse_loss(x, xrec) + kl_weight * kl_loss, moments, z, xrec, kl_loss


def vae_decode(sd: SD, dd: dict, z: Tensor) -> Tensor:
    """Decoder.forward, modules/diffusion/model.py:707-765: post_quant_conv (standalone, :700-704) -> conv_in -> mid ->
    for level = L-1 .. 0: (num_res_blocks + 1) resnets (+ attn) then Upsample (nearest x2 + 3x3 conv, :44-62) except at
    level 0 -> GroupNorm, SiLU, conv_out (tanh_out / give_pre_end are off in the SD/SDXL configs)."""
    levels = len(dd["ch_mult"])
    h = conv(sd, "post_quant_conv", z, padding=0) if "post_quant_conv.weight" in sd else z
    h = conv(sd, "conv_in", h)
    h = vae_resnet(sd, "mid.block_1", h)
    h = vae_attn(sd, "mid.attn_1", h)
    h = vae_resnet(sd, "mid.block_2", h)
    for lvl in reversed(range(levels)):
        for ib in range(dd["num_res_blocks"] + 1):
            h = vae_resnet(sd, f"up.{lvl}.block.{ib}", h)
            if f"up.{lvl}.attn.{ib}.norm.weight" in sd:
                h = vae_attn(sd, f"up.{lvl}.attn.{ib}", h)
        if lvl != 0:
            h = conv(sd, f"up.{lvl}.upsample.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"))
    return conv(sd, "conv_out", F.silu(group_norm(sd, "norm_out", h, 1e-6)))


# ------------------------------------------------------------------------------------------------
# diffusion glue
# ------------------------------------------------------------------------------------------------
def legacy_ddpm_sigmas(num_idx: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120) -> Tensor:
    """LegacyDDPMDiscretization + Discretization.__call__, modules/diffusion/discretization.py:17-36,149-171 with
    make_beta_schedule("linear") (util.py): betas = linspace(sqrt(s), sqrt(e), n, float64)**2; sigmas ascending->flipped;
    `do_append_zero` of the call is ignored and the instance default True is used, flip=False:
    the table handed to DiscreteDenoiser is [sigma_max ... sigma_min, 0.0] (1001 entries)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_idx, dtype=torch.float64) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0, dtype=torch.float32)  # fp32 cumprod, discretization.py:159
    sigmas = ((1 - alphas_cumprod) / alphas_cumprod) ** 0.5
    sigmas = torch.flip(sigmas, (0,)).to(torch.float32)
    return torch.cat([sigmas, sigmas.new_zeros([1])])


def legacy_ddpm_sampling_sigmas(n: int, num_idx: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.0120) -> Tensor:
    """The sampler's table, LegacyDDPMDiscretization.get_sigmas for n < num_timesteps (discretization.py:13-14,161-171):
    n roughly equally spaced timesteps ending at num_idx - 1, descending sigmas, final 0 appended."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_idx, dtype=torch.float64) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0, dtype=torch.float32)
    if n < num_idx:
        import numpy as np

        steps = np.linspace(num_idx - 1, 0, n, endpoint=False).astype(int)[::-1]
        alphas_cumprod = alphas_cumprod[torch.from_numpy(np.ascontiguousarray(steps))]
    sigmas = torch.flip(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, (0,)).to(torch.float32)
    return torch.cat([sigmas, sigmas.new_zeros([1])])


def sigma_to_idx(table: Tensor, sigma: Tensor) -> Tensor:
    """DiscreteDenoiser.sigma_to_idx, modules/diffusion/denoiser.py:83-85."""
    dists = sigma - table[:, None]
    return dists.abs().argmin(dim=0).view(sigma.shape)


def eps_denoiser(net, table: Tensor, z: Tensor, sigma: Tensor) -> Tensor:
    """DiscreteDenoiser.forward with EpsPreconditioning, modules/diffusion/denoiser.py:28-57,87-97 and
    denoiser_preconditioning.py:33-44: sigma snapped to the table; c_skip=1, c_out=-sigma, c_in=(sigma^2+1)^-1/2,
    c_noise = table index of sigma."""
    sigma = table[sigma_to_idx(table, sigma)]
    s = sigma[:, None, None, None]
    c_in = 1.0 / (s ** 2.0 + 1.0) ** 0.5
    c_noise = sigma_to_idx(table, sigma.clone())
    out = net(z * c_in, c_noise)
    return out * (-s) + z * 1.0


def edm_loss(net, table: Tensor, x: Tensor, sigma: Tensor, noise: Tensor) -> Tensor:
    """StandardDiffusionLoss._forward 'edm' branch with injected sigma / noise, modules/diffusion/loss.py:117-157;
    EpsWeighting sigma^-2 (denoiser_weighting.py:22-25); BatchMSELoss = per-sample mean (losses/functions.py:81-94).
    The weight uses the UN-snapped sigma, exactly as loss.py:144 does.  Returns loss[B]."""
    z = x + sigma[:, None, None, None] * noise
    d = eps_denoiser(net, table, z, sigma)
    w = sigma ** -2.0
    return ((d.float() - x.float()) ** 2).flatten(1).mean(1) * w.float()


def rf_xl_coefficients(sigma: Tensor):
    """RectifiedFlowXLPreconditioning, modules/diffusion/denoiser_preconditioning.py:77-90: (c_skip, c_out, c_in, c_noise)."""
    c_in = (1.0 / (1.0 + sigma)) / ((1.0 / (sigma + 1.0)) ** 2.0 + (sigma / (sigma + 1.0)) ** 2.0) ** 0.5
    return torch.ones_like(sigma), -sigma, c_in, 1000.0 * (sigma / (1 + sigma))


def rf_weighting(sigma: Tensor, m: float = 0.0, s: float = 1.0) -> Tensor:
    """RectifiedFlowWeighting.__call__, modules/diffusion/denoiser_weighting.py:38-55 (fp64, as the reference computes it)."""
    sigma = sigma.to(torch.float64)
    t = sigma / (1.0 + sigma)
    cfm = 1 / (1 - t) ** 2
    half_pi = torch.acos(torch.zeros(1, dtype=torch.float64))[0]
    pi_w = (1 / (s * (4.0 * half_pi) ** 0.5)) * (1 / (t * (1.0 - t))) * torch.exp(-0.5 * (torch.log(sigma) - m) ** 2 / s ** 2)
    return cfm * pi_w


def noise_with_offset(noise: Tensor, offset: Optional[Tensor], noise_offset: float) -> Tensor:
    """DiffusionLoss.apply_noise_offset with the drawn per-(sample, channel) `offset` handed in, modules/diffusion/loss.py:32-40
    (noise_offset clamped to [0, 1] by the constructor, :27-30)."""
    noise_offset = min(max(noise_offset, 0.0), 1.0)
    if noise_offset <= 0 or offset is None:
        return noise
    return noise + noise_offset * offset.to(noise)


def diffusion_loss(net, table: Tensor, x: Tensor, sigma: Tensor, noise: Tensor, loss_type: str = "l2", objective: str = "edm") -> Tensor:
    """StandardDiffusionLoss._forward + get_loss with injected sigma / noise, modules/diffusion/loss.py:105-157, both objectives
    and both loss types (BatchMSELoss / BatchL1Loss with reduction "mean" = per-sample mean, losses/functions.py:65-94).
      edm: z = x + sigma*noise; D = DiscreteDenoiser(EpsPreconditioning)(z) ("D" output); loss = l(D, x) * EpsWeighting(sigma)
      rf : z = (1-sigma)*x + sigma*noise; F = Denoiser(RectifiedFlowXLPreconditioning) "F" output = raw network output on
           (z*c_in, c_noise); loss = l(F, noise) * RectifiedFlowWeighting(sigma)                        (loss.py:126-137)
    Returns loss[B]."""
    sb = sigma[:, None, None, None]
    if objective == "rf":
        z = (1.0 - sb) * x + sb * noise
        _, _, c_in, c_noise = rf_xl_coefficients(sb)
        out = net(z * c_in.to(z.dtype), c_noise.reshape(sigma.shape))
        target, w = noise, rf_weighting(sigma)
    else:
        z = x + sb * noise
        out = eps_denoiser(net, table, z, sigma)
        target, w = x, sigma ** -2.0
    diff = out.float() - target.float()
    per = diff * diff if loss_type == "l2" else diff.abs()
    return per.flatten(1).mean(1) * w.float()


def training_step_loss(unet_sd: SD, unet_cfg: dict, vae_sd: SD, vae_dd: dict, scale_factor: float, image: Tensor,
                       sigma: Tensor, noise: Tensor, context: Tensor, y: Optional[Tensor], table: Optional[Tensor] = None):
    """DiffusionEngine.training_step, models/diffusion.py:205-233 with encode_first_stage :186-197 and
    OpenAIWrapper.forward (wrappers.py:25-40, no concat cond): returns (loss.mean(), loss[B], latents)."""
    if table is None:
        table = legacy_ddpm_sigmas()
    with torch.no_grad():
        latents = scale_factor * vae_encode(vae_sd, vae_dd, image)

    def net(xin, t):
        return unet_forward(unet_sd, unet_cfg, xin, t, context, y)

    loss = edm_loss(net, table, latents, sigma, noise)
    return loss.mean(), loss, latents
