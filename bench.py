#!/usr/bin/env python
"""bench.py -- SDXL training-step throughput on MI355X (BASELINE.json metric: train images/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: starts N fresh rank processes itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): SDXL-base UNet (2.567 B parameters, random init), 1024x1024 synthetic
images, batch 4 per GPU, bf16 MFMA arithmetic with fp32 accumulation / statistics / master weights; frozen VAE
encoder and the frozen SDXL conditioner (CLIP ViT-L + OpenCLIP ViT-bigG text encoders on synthetic token ids -- the
CLIP vocabulary is not on the box -- plus the size / crop embedders) in the step, as in the reference's
training_step (--precomputed-te feeds synthetic text-encoder OUTPUTS instead).  One "step" = VAE encode ->
conditioner -> noise + preconditioning -> UNet forward -> weighted MSE -> UNet backward -> (N>1: flat gradient
all-reduce over RCCL, overlapped with backward) -> fused Adafactor (or AdamW) update of every UNet parameter.
Nothing is skipped or cached inside the timed region.  Weak scaling: every rank processes its own 4 images.

Prints ONE JSON line on rank 0 (see the task contract); extra objects:
  roofline     -- the dominant kernel (the MFMA tile engine: nk_gemm_*_kernel<...> and nk_conv3x3_halo_kernel<...>, ~88 % of the step's FLOPs):
                  algorithmic FLOPs of every launch (2*M*N*K) / its duration from HIP events recorded around each
                  launch on the launch stream during an instrumented replay of the same step.
  cpu_baseline -- the CPU oracle's training step on the host cores (bounded sample, see `sample`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SDXL_UNET = dict(
    adm_in_channels=2816, num_classes="sequential", use_checkpoint=False, in_channels=4, out_channels=4, model_channels=320,
    attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64, use_linear_in_transformer=True,
    transformer_depth=[1, 2, 10], context_dim=2048, spatial_transformer_attn_type="softmax-xformers",
)
SDXL_VAE_DD = dict(attn_type="vanilla-xformers", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                   num_res_blocks=2, attn_resolutions=[], dropout=0.0)
SCALE_FACTOR = 0.13025
TFLOP_PER_IMAGE = 25.16  # UNet fwd+bwd 20.284 + VAE encode 4.879 (algorithmic, no recompute)
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def reinit_zero_modules(net, std=0.02, seed=0):
    """zero_module()-initialised layers get N(0, std^2) so that gradients are non-trivial."""
    g = torch.Generator(device=next(net.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g, device=p.device) * std)


def build_conditioner(device):
    """configs/sdxl/sdxl.example.yaml:118-160: CLIP-L hidden layer 11, OpenCLIP bigG penultimate + pooled, three size embedders"""
    from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2
    from neurosis_amd.modules.encoders import ConcatTimestepEmbedderND, GeneralConditioner

    torch.manual_seed(7)
    embedders = [FrozenCLIPEmbedder(layer="hidden", layer_idx=11, input_key="caption_ids", device=device),
                 FrozenOpenCLIPEmbedder2(arch="ViT-bigG-14", version=None, layer="penultimate", always_return_pooled=True, legacy=False,
                                         input_key="caption_ids", device=device)]
    embedders += [ConcatTimestepEmbedderND(outdim=256, input_key=k) for k in ("original_size_as_tuple", "crop_coords_top_left", "target_size_as_tuple")]
    return GeneralConditioner(embedders).to(device)


def build_engine(device, image_hw=(1024, 1024), conditioner=None):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models.autoencoder import AutoencoderKL
    from neurosis_amd.models.diffusion import DiffusionEngine

    torch.manual_seed(42)
    with torch.device(device):
        unet = D.UNetModel(**SDXL_UNET)
        vae = AutoencoderKL(embed_dim=4, ddconfig=SDXL_VAE_DD)
        denoiser = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    denoiser = denoiser.to(device)
    reinit_zero_modules(unet)
    loss_fn = D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting())
    eng = DiffusionEngine(model=unet, denoiser=denoiser, first_stage_model=vae, loss_fn=loss_fn, scale_factor=SCALE_FACTOR, input_key="image",
                          conditioner=conditioner)
    eng.setup_flat_params()
    return eng


MIXED_BUCKETS = [(1216, 832), (832, 1216), (1024, 1024), (1152, 896), (896, 1152)]   # (H, W)


def synthetic_batch(device, batch, hw, gen, precomputed_te=True):
    H, W = hw
    image = torch.rand(batch, 3, H, W, device=device, generator=gen) * 2 - 1
    if precomputed_te:
        return {"image": image, "crossattn": torch.randn(batch, 77, 2048, device=device, generator=gen),
                "vector": torch.randn(batch, 2816, device=device, generator=gen)}
    # token ids in the CLIP convention: BOS, a prompt of random length, EOS (the highest id) to the end of the context
    ids = torch.randint(1000, 40000, (batch, 77), device=device, generator=gen)
    length = torch.randint(5, 76, (batch, 1), device=device, generator=gen)
    ids = torch.where(torch.arange(77, device=device)[None] >= length, torch.full_like(ids, 49407), ids)
    ids[:, 0] = 49406
    size = _size_tensor(device, batch, H, W)
    return {"image": image, "caption_ids": ids, "original_size_as_tuple": size, "crop_coords_top_left": torch.zeros_like(size), "target_size_as_tuple": size}


_SIZE_CACHE = {}


def _size_tensor(device, batch, H, W):
    """[batch, 2] fp32 (H, W) on the device, built once per bucket: a fresh torch.tensor(list, device=...) is a pageable host-to-device copy,
    which makes the HOST wait for everything queued on the stream -- once per step that cost the launching thread its whole lead over the
    GPU (tools/sync_debug.py; round 4)."""
    key = (str(device), batch, H, W)
    if key not in _SIZE_CACHE:
        _SIZE_CACHE[key] = torch.tensor([[float(H), float(W)]] * batch, device=device)
    return _SIZE_CACHE[key]


def draw_sigmas(batch, gen, device):
    """sigma = exp(-1.2 + 1.2 n), clipped to the LegacyDDPM table range (the denoiser snaps it).  Drawn ON the device: a CPU
    draw copied over is a synchronising host-to-device copy (see _size_tensor)."""
    n = torch.randn(batch, device=device, generator=gen)
    return (-1.2 + 1.2 * n).exp().clamp(0.0292, 14.61)


class GemmTimer:
    """HIP events around every launch of the MFMA tile engine (all nk_linear_* / nk_conv2d_* entry points), on the
    stream the kernels are launched on; algorithmic FLOPs = 2*M*N*K per launch."""

    NAMES = ("nk_linear_fwd", "nk_linear_dgrad", "nk_linear_wgrad", "nk_conv2d_fwd", "nk_conv2d_dgrad", "nk_conv2d_wgrad",
             "nk_linear_fwd_batched", "nk_linear_wgrad_batched", "nk_conv2d_fwd_stats", "nk_conv2d_dgrad_flipped", "nk_linear_wgrad_bias",
             "nk_conv2d_wgrad_bias", "nk_linear_dgrad_geglu", "nk_linear_fwd_geglu", "nk_linear_dgrad_geglu_s", "nk_linear_fwd_geglu_s")
    # the batched entry points (several same-shape GEMMs per launch) are reported with the family they belong to; so are the
    # convolutions with a fused GroupNorm statistics epilogue and the input gradients that run as forward convolutions of dy
    FAMILY = {"nk_linear_fwd_batched": "nk_linear_fwd", "nk_linear_wgrad_batched": "nk_linear_wgrad", "nk_conv2d_fwd_stats": "nk_conv2d_fwd",
              "nk_conv2d_dgrad_flipped": "nk_conv2d_dgrad", "nk_linear_wgrad_bias": "nk_linear_wgrad", "nk_conv2d_wgrad_bias": "nk_conv2d_wgrad",
              "nk_linear_dgrad_geglu": "nk_linear_dgrad", "nk_linear_fwd_geglu": "nk_linear_fwd",
              "nk_linear_dgrad_geglu_s": "nk_linear_dgrad", "nk_linear_fwd_geglu_s": "nk_linear_fwd"}     # (round 6: the saved-derivative forms, same arguments)

    def __init__(self):
        self.records = []
        self.tag = ""          # appended to the family name of launches made while it is set ("[frozen VAE]": the encoder's convolutions)

    @staticmethod
    def flops(name, args):
        if name == "nk_linear_fwd":
            return 2.0 * args[5] * args[6] * args[7]
        if name in ("nk_linear_dgrad",):
            return 2.0 * args[4] * args[5] * args[6]
        if name == "nk_linear_wgrad":
            return 2.0 * args[3] * args[4] * args[5]
        if name in ("nk_linear_wgrad_bias", "nk_linear_dgrad_geglu", "nk_linear_dgrad_geglu_s"):        # (dy, x | w, dw | u, dbias | du, M, N, K | I, ...)
            return 2.0 * args[4] * args[5] * args[6]
        if name in ("nk_linear_fwd_geglu", "nk_linear_fwd_geglu_s"):       # (x, w, bias, u | s, h, M, I, K, ...): the GEMM is M x 2I x K
            return 2.0 * args[5] * 2 * args[6] * args[7]
        if name == "nk_linear_fwd_batched":     # (ptrs, ptrs, ptrs, count, M, N, K, ...)
            return 2.0 * args[3] * args[4] * args[5] * args[6]
        if name == "nk_linear_wgrad_batched":   # (ptrs, ptrs, ptrs, bias ptrs, count, M, N, K, ...)
            return 2.0 * args[4] * args[5] * args[6] * args[7]
        d = args[0]._obj
        up = 2 if d.upsample else 1
        if name in ("nk_conv2d_dgrad", "nk_conv2d_dgrad_flipped"):  # rows = input pixels (virtual 2x grid when upsampling), algorithmic = same MACs as fwd
            return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin
        return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.KH * d.KW * d.Cin

    def install(self):
        from neurosis_amd import lib, ops

        self._orig = lib.call
        timer = self

        def timed_call(name, *args):
            if name not in timer.NAMES:
                return timer._orig(name, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            timer._orig(name, *args)
            e.record()
            timer.records.append((name + timer.tag, timer.flops(name, args), s, e))

        lib.call = timed_call
        ops.call = timed_call
        import neurosis_amd.nn as nkn
        import neurosis_amd.modules.diffusion.loss as nkl

        self._mods = [(ops, "call"), (nkn, "call"), (nkl, "call")]
        for m, a in self._mods[1:]:
            setattr(m, a, timed_call)

    def uninstall(self):
        from neurosis_amd import lib

        lib.call = self._orig
        for m, a in self._mods:
            setattr(m, a, self._orig)

    @staticmethod
    def event_overhead_ms() -> float:
        """What an event pair with NOTHING between its two records reads on this stream (median of 200): the timestamp writes
        themselves.  Subtracted from every launch interval, so that a launch's figure is the kernel's, as rocprofv3 reports it."""
        torch.cuda.synchronize()
        pairs = []
        for _ in range(200):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        v = sorted(s.elapsed_time(e) for s, e in pairs)
        return v[len(v) // 2]

    def summary(self):
        torch.cuda.synchronize()
        self.overhead_ms = self.event_overhead_ms()
        tot_f = tot_ms = 0.0
        per = {}
        for name, f, s, e in self.records:
            ms = max(s.elapsed_time(e) - self.overhead_ms, 1e-4)
            tot_f += f
            tot_ms += ms
            base, _, tag = name.partition("[")
            a = per.setdefault(self.FAMILY.get(base, base) + ("[" + tag if tag else ""), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += f
            a[2] += ms
        n = max(len(self.records), 1)
        return tot_f, tot_ms, n, {k: {"launches": v[0], "tflops": v[1] / max(v[2], 1e-9) / 1e9, "ms": v[2]} for k, v in per.items()}


def host_cores() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one (a GPU box exposes every host core in
    sched_getaffinity but caps the container at its share), else the affinity mask."""
    return host_cores_and_source()[0]


def host_cores_and_source():
    """(threads, where the number comes from).  Without a cgroup quota a pool box shows every host core in the affinity mask although the
    container's share is 16 per GPU (the pool's documented rule): that case, and only that one, falls back to the rule."""
    n, src = os.cpu_count() or 1, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, src = a, "sched_getaffinity"
    except Exception:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period) + 0.999))
            if q <= n:
                return q, "cgroup cpu.max quota"
    except Exception:
        pass
    if n > 16:
        return 16, f"pool rule: 16 CPUs per GPU share ({src} shows {n}, no cgroup quota)"
    return max(1, n), src


SD15_UNET = dict(use_checkpoint=False, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1], num_res_blocks=2,
                 channel_mult=[1, 2, 4, 4], num_heads=8, transformer_depth=1, context_dim=768, spatial_transformer_attn_type="softmax-xformers")
TFLOP_PER_IMAGE_SD15 = 3.53  # SD1.5 512^2, UNet fwd+bwd 2.410 + VAE encode 1.117


def cpu_baseline():
    """The CPU oracle's training step (fp32, torch CPU eager) on the host cores, on a BOUNDED sample: BASELINE.json
    configs[0] -- SD1.5 512x512, batch 1 (the reference's own CPU-runnable case, BASELINE.md section 4: the reference took
    11.2 s/step for it on 8 vCPU) -- one full step: VAE encode + UNet forward + loss + backward.  3.53 algorithmic TFLOP
    against 25.16 per SDXL 1024^2 image, so value = (1 / seconds) * 3.53 / 25.16 SDXL-1024^2-image equivalents per second.
    (A full-width SDXL 1024^2 step on the CPU would take minutes; a down-sized SDXL sample is dominated by streaming the
    10 GB of weights and under-reports the CPU.)  Weights are constant-filled: values do not change the arithmetic cost."""
    t_start = time.time()
    try:
        from oracle import sdxl_oracle as O
        import neurosis_amd.modules.diffusion as D
        from neurosis_amd.models.autoencoder import AutoencoderKL

        cores = host_cores()
        torch.set_num_threads(cores)
        with torch.device("meta"):
            unet = D.UNetModel(**SD15_UNET)
            vae = AutoencoderKL(embed_dim=4, ddconfig=SDXL_VAE_DD)

        def mk(sd, grad):
            out = {}
            for k, v in sd.items():
                fan = max(v[0].numel(), 1) if v.dim() > 1 else 1
                val = 1.0 if (v.dim() == 1 and k.endswith("weight")) else (0.0 if v.dim() == 1 else 0.5 * fan ** -0.5)
                out[k] = torch.full(v.shape, val, dtype=torch.float32).requires_grad_(grad)
            return out

        usd = mk(unet.state_dict(), True)
        vsd = mk({**vae.encoder.state_dict(), **{f"quant_conv.{k}": v for k, v in vae.quant_conv.state_dict().items()}}, False)
        print(f"[cpu_baseline] oracle weights ready after {time.time() - t_start:.1f} s, {cores} threads", file=sys.stderr, flush=True)
        g = torch.Generator().manual_seed(0)
        img = torch.rand(1, 3, 512, 512, generator=g) * 2 - 1
        ctx = torch.randn(1, 77, 768, generator=g)
        sigma = torch.tensor([1.0])
        noise = torch.randn(1, 4, 64, 64, generator=g)
        t1 = time.time()
        loss, _, _ = O.training_step_loss(usd, dict(SD15_UNET), vsd, SDXL_VAE_DD, 0.18215, img, sigma, noise, ctx, None)
        loss.backward()
        t = time.time() - t1
        print(f"[cpu_baseline] SD1.5 512^2 step: {t:.2f} s", file=sys.stderr, flush=True)
        return {"value": round((1.0 / t) * TFLOP_PER_IMAGE_SD15 / TFLOP_PER_IMAGE, 6), "unit": "images/s", "cores": cores, "cores_source": host_cores_and_source()[1],
                "kind": "port", "sample": f"oracle (torch CPU fp32 eager), BASELINE config 1: one SD1.5 512x512 batch-1 training step (VAE encode + UNet fwd + loss + bwd, "
                          f"3.53 algorithmic TFLOP) in {t:.2f} s = {TFLOP_PER_IMAGE_SD15 / t:.3f} TFLOP/s; value = SDXL-1024^2-image equivalents/s "
                          f"(x 3.53/25.16)"}
    except Exception as ex:  # the baseline is reported, never required
        return {"value": None, "unit": "images/s", "cores": host_cores(), "cores_source": host_cores_and_source()[1], "kind": "port",
                "sample": f"failed: {type(ex).__name__}: {ex}"}


def pmc_traffic():
    """HBM bytes per tile-engine launch from the committed PMC profile of this same command (rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; both are KB)."""
    import csv

    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((p for p in (os.path.join(prof, f"r{r:02d}_pmc_summary.csv") for r in range(9, 0, -1)) if os.path.exists(p)), "")      # newest round
    try:
        rows = list(csv.reader(open(path)))[1:]
    except OSError:
        return None, None
    n = b = 0.0
    for r in rows:
        if r[0].startswith(("void nk_gemm", "nk_gemm", "void nk_conv3x3_halo")):      # the tile engine's kernel families
            n += float(r[1])
            b += float(r[1]) * (float(r[2]) + float(r[3]))
    return (round(b / n), "profiles/" + os.path.basename(path)) if n else (None, None)


def expected_exchange_ms(grad_elems: int, world: int, link_GBps: float = 153.0) -> dict:
    """SURVEY section 5's price list for the gradient exchange on one node: 8 GPUs fully connected, 7 xGMI links x ~153 GB/s per GPU.
    A ring all-reduce is bound by ONE link (2 (N-1)/N x S over it); a direct reduce-scatter + all-gather uses the N-1 links at once
    (2 S / N per link); rs_ag sends fp32 gradients out and bf16 shadows back (S/N + S/2N per link)."""
    if world < 2:
        return {}
    out = {}
    for name, nbytes in (("fp32", grad_elems * 4), ("bf16", grad_elems * 2)):
        out[f"ring_allreduce_{name}"] = round(2 * (world - 1) / world * nbytes / (link_GBps * 1e9) * 1e3, 1)
        out[f"direct_rs_ag_{name}"] = round(2 * nbytes / world / (link_GBps * 1e9) * 1e3, 1)
    out["direct_rs_fp32_ag_bf16"] = round(1.5 * grad_elems * 4 / world / (link_GBps * 1e9) * 1e3, 1)
    out["assumes"] = f"{link_GBps:.0f} GB/s per xGMI link, {world - 1} links per GPU, no overlap with backward"
    return out


def launch_ranks(args, argv, runner=None):
    """`python bench.py --gpus N` from a bare shell (N > 1, no WORLD_SIZE in the environment): start N FRESH child processes, one rank per
    GPU, through torch.distributed.run on 127.0.0.1 and return their exit code; rank 0's JSON line reaches stdout through the inherited pipe.
    Runs BEFORE this process makes any GPU call (a process that has initialised HIP must never be replaced or forked into ranks), and the
    children are ordinary subprocesses, not an exec.  Returns None when this process is itself a rank (or N = 1).
    Replaces the Lightning launcher of the reference's `trainer: devices: N` (/root/reference/configs/sdxl/sdxl.example.yaml:3-15)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    assert not torch.cuda.is_initialized(), "the rank launcher must run before any GPU call"
    import socket
    import subprocess

    with socket.socket() as s:      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return (runner or subprocess.run)(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--accumulate", type=int, default=1,
                    help="micro-batches per optimizer step (the reference example config uses accumulate_grad_batches: 4); "
                         "a 'step' then is one optimizer step = ACCUMULATE micro-batches of --batch images; not the metric's configuration")
    ap.add_argument("--mixed-res", action="store_true",
                    help="BASELINE config 4: every step each rank draws one aspect bucket (W,H) from the SDXL bucket list "
                         "{832x1216, 1216x832, 1024x1024, 896x1152, 1152x896}; not the metric's configuration")
    ap.add_argument("--optimizer", default="adafactor", choices=["adafactor", "adamw"],
                    help="adafactor = the reference example config's optimizer (scale_parameter, relative_step, warmup_init); adamw = fused flat AdamW")
    ap.add_argument("--precomputed-te", action="store_true",
                    help="feed synthetic text-encoder outputs instead of running the frozen conditioner (CLIP-L + OpenCLIP-bigG) in the step")
    ap.add_argument("--serialize", action="store_true",
                    help="run EVERY step with the weight-gradient side stream and the optimizer stream disabled (one kernel at a time): the command whose "
                         "rocprofv3 kernel trace is profiles/r02_gemm_serialized_kernel_stats.csv; not the metric's configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for rehearsing the multi-rank control flow)")
    ap.add_argument("--wire-dtype", choices=["fp32", "bf16"], default="fp32",
                    help="N > 1: dtype of the gradient all-reduce (fp32 = what Lightning DDP exchanges for the reference's fp32 parameters; "
                         "bf16 halves the bytes on xGMI and sums in bf16)")
    ap.add_argument("--recompute", choices=["none", "norms"], default="none",
                    help="selective activation recompute in the transformer blocks (UNetModel.set_recompute): norms = LayerNorm outputs and GEGLU products "
                         "rebuilt in backward instead of held (less memory, two more elementwise passes per block); reported in config")
    ap.add_argument("--stream-optimizer", action="store_true",
                    help="N = 1 A/B: issue each top-level UNet block's Adafactor update behind that block's backward (DiffusionEngine.stream_optimizer) "
                         "instead of the whole update after backward")
    ap.add_argument("--dp-mode", choices=["allreduce", "rs_ag"], default=os.environ.get("NK_DP_MODE", "allreduce"),
                    help="N > 1: allreduce = flat all-reduce of every gradient slice, the whole optimizer on every rank (default, what Lightning DDP does); "
                         "rs_ag = every slice reduce-scattered into tensor-aligned parts, optimizer on the owned parts, bf16 shadows all-gathered (neurosis_amd/dp.py)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--alt-timeout", type=float, default=180.0, help="seconds one alternative exchange configuration may take before the line is printed without it")
    ap.add_argument("--alt-steps", type=int, default=5,
                    help="N > 1: behind the timed steps, run this many steps in each of the OTHER exchange configurations (dp-mode x wire dtype) in the "
                         "same process and report them as comm.alt (0 = off)")
    return ap.parse_args(argv)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    rc = launch_ranks(args, argv)
    if rc is not None:
        raise SystemExit(rc)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one process per GPU (python bench.py --gpus N does it itself)")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist

    # NK_DP_FORCE=1 at N = 1: the single rank issues every collective of the exchange all the same (neurosis_amd/dp.py) -- the one way a 1-GPU box
    # can run the full-size step's exchange through RCCL itself; what it measures is the cost of the exchange machinery without any wire time
    forced = world == 1 and os.environ.get("NK_DP_FORCE", "0") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or forced:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from neurosis_amd import lib
    from neurosis_amd.dp import FlatDataParallel

    lib.load()  # fail loudly if the HIP library is missing
    eng = build_engine(device, (args.res, args.res), None if args.precomputed_te else build_conditioner(device))
    unet = eng.model.diffusion_model
    if args.recompute != "none":
        unet.set_recompute(args.recompute)
    if args.stream_optimizer:
        eng.stream_optimizer = True
    if args.optimizer == "adafactor":
        eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)   # configs/sdxl/sdxl.example.yaml:158-164
    dp = FlatDataParallel(unet, eng.store, wire_dtype=torch.bfloat16 if args.wire_dtype == "bf16" else None, mode=args.dp_mode) if world > 1 or forced else None
    if dp is not None and dp.sharded:
        if args.optimizer != "adafactor":
            raise SystemExit("--dp-mode rs_ag needs --optimizer adafactor (the chunked fused optimizer)")
        dp.attach_optimizer(eng.adafactor)
    if args.serialize:
        eng.store.state.wgrad_stream = None
        eng.overlap_optimizer = False
    gen = torch.Generator(device=device).manual_seed(42 + rank)
    gen_cpu = torch.Generator().manual_seed(42 + rank)

    step_marks = []    # one event per timed step start (+ one at the end): step-time percentiles without host syncs
    comm_marks = []    # N > 1: (backward done, exchange joined, first collective start, last collective end) per step

    def step(mark=False, force_hw=None):
        if mark:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            step_marks.append(ev)
        for mb in range(args.accumulate):
            hw = (args.res, args.res)
            if args.mixed_res:   # (H, W) of this rank's bucket for this step (N/dataset/aspect/lists.py:14-56)
                hw = MIXED_BUCKETS[int(torch.randint(len(MIXED_BUCKETS), (1,), generator=gen_cpu))]
            if force_hw is not None:
                hw = force_hw
            batch = synthetic_batch(device, args.batch, hw, gen, args.precomputed_te)
            sig = draw_sigmas(args.batch, gen, device)
            eng.accumulate(mb, dp, last=mb == args.accumulate - 1)
            loss = eng.training_step(batch, 0, sigmas=sig)
            (loss / args.accumulate).backward()
        gs = 1.0
        if dp is not None:
            if mark:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
            gs = dp.finish()
            if mark:
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record()
                comm_marks.append((e1, e2) + dp.reducer.take_timing())
        eng.optimizer_step(lr=1e-6, weight_decay=1e-2, grad_scale=gs, dp=dp)
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, before the W warm-up steps: the UNet chain is replayed from hipGraphs (neurosis_amd/graphs.py), which are captured on
    # the SECOND step of each input signature -- two untimed priming steps per resolution, so that no capture (a device
    # synchronisation and ~1 s of host work) can land in the warm-up-then-timed region whatever W is.
    from neurosis_amd.graphs import graphs_enabled
    priming = 0
    if graphs_enabled("unet"):
        for hw in (MIXED_BUCKETS if args.mixed_res else [(args.res, args.res)]):
            for _ in range(2):
                step(force_hw=hw)
                priming += 1
    for _ in range(args.warmup):
        step()
    # The long-lived object graph (modules, parameters, descriptor tables) goes to the permanent generation: a full
    # collection of the cyclic GC otherwise walks all of it every ~10 steps and holds the launching thread for ~70 ms
    # (measured: one 268 ms step in twelve).  Garbage made by the steps themselves is still collected.
    import gc
    gc.collect()
    gc.freeze()
    if dp is not None:
        dp.reducer.record_timing = True
        dp.reducer.take_counts()       # count the timed steps' collectives only
    barrier()
    if os.environ.get("NK_SYNC_DEBUG_AFTER_WARMUP") == "1":      # tools/sync_debug.py: report every host-synchronising call of the timed steps
        torch.cuda.set_sync_debug_mode("warn")
    setup_peak = torch.cuda.max_memory_allocated()      # priming + warm-up: an eager pass and the graph captures live side by side
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step(mark=True)
    end_mark = torch.cuda.Event(enable_timing=True)
    end_mark.record()
    torch.cuda.set_sync_debug_mode("default")
    barrier()
    dt = time.perf_counter() - t0
    if dp is not None:
        dp.reducer.record_timing = False
    # per-step GPU times (event to event on the compute stream), this rank
    marks = step_marks + [end_mark]
    in_order = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(marks) - 1)]
    per_step = sorted(in_order)
    pct = lambda q: per_step[min(len(per_step) - 1, int(round(q * (len(per_step) - 1))))]
    def comm_summary(dp_, marks_, wire_dtype):
        """what the gradient exchange of the steps just run did: collectives, bytes each rank sent, exposed time, bus bandwidth"""
        exposed = [a.elapsed_time(b) for a, b, _, _ in marks_]
        span = [f.elapsed_time(l) for _, _, f, l in marks_ if f is not None and l is not None]
        ncoll, wire = dp_.reducer.take_counts()
        nsteps_counted = max(len(marks_), 1)
        sent = int(wire / nsteps_counted)
        span_s = (sum(span) / len(span) * 1e-3) if span else None
        return {"mode": dp_.mode, "wire_dtype": wire_dtype, "collectives_per_step": round(ncoll / nsteps_counted, 1),
                "bytes_sent_per_rank_per_step": sent, "exposed_ms_mean": round(sum(exposed) / len(exposed), 2),
                "first_to_last_collective_ms_mean": round(sum(span) / max(len(span), 1), 2) if span else None,
                # bus bandwidth from the bytes the collectives actually moved (all-reduce: 2 (N-1)/N x payload; rs_ag: (N-1)/N x the padded
                # staging rows of the reduce-scatters -- the all-gathers run behind the update, outside this span)
                "busbw_GBps_over_span": round(sent / span_s / 1e9, 1) if span_s else None}

    comm = None
    if comm_marks:
        nbytes = eng.store.grad.numel() * (2 if args.wire_dtype == "bf16" else eng.store.grad.element_size())
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:  # noqa: BLE001 - version query only
            rccl = None
        devices = [None] * world
        dist.all_gather_object(devices, f"{torch.cuda.get_device_name(device)} #{torch.cuda.current_device()} pid {os.getpid()}")
        comm = {"world_size": world, "backend": dist.get_backend(), "rccl_version": rccl, "devices": devices}
        comm.update(comm_summary(dp, comm_marks, args.wire_dtype))
        comm.update({"gradient_bytes": nbytes, "max_chunk_elements": dp.reducer.max_chunk,
                     "nccl_env": {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_NCHANNELS_PER_NET_PEER", "RCCL_MSCCL_ENABLE") if k in os.environ},
                     "expected_ms": expected_exchange_ms(eng.store.grad.numel(), world),
                     "note": "exposed = compute stream stalled between end of backward and end of the gradient exchange; span = first to last collective of the gradient exchange (includes the backward it overlaps); rs_ag counts its all-gathers in bytes_sent but they run on the optimizer stream; alt = the OTHER exchange configurations run for a few steps in this same process behind the timed ones (ms_per_step = max over ranks, barrier to barrier), so that one multi-GPU run prices every mode"})
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_val = float(last.detach())
    images = args.steps * args.batch * world * args.accumulate
    value = images / dt
    ms_per_step = dt / args.steps * 1e3

    roofline = None
    if not args.no_roofline:
        # Two instrumented replays of the same step (every rank: the step contains the gradient all-reduce, a collective; only
        # rank 0 reports), HIP events around every tile-engine launch on the stream it is launched on:
        #   in step     -- exactly as timed above: dgrad chain and weight gradients share the chip on two streams, so a launch's
        #                  duration includes what the other stream costs it.  This is `achieved` / `frac`: it is what the rocprofv3
        #                  kernel trace of this same command shows (profiles/r02_bench_kernel_stats.csv).
        #   serialized  -- side stream off, one kernel at a time: the kernels' own speed (`achieved_serialized` / `frac_serialized`;
        #                  trace of `bench.py --serialize`: profiles/r02_gemm_serialized_kernel_stats.csv).
        est = eng.store.state

        def replay(serialized: bool):
            timer = GemmTimer()
            timer.install()
            side = est.wgrad_stream
            if serialized:
                est.wgrad_stream = None
            graph_env = os.environ.get("NK_GRAPH")
            os.environ["NK_GRAPH"] = "0"      # the timer wraps the Python-side launches: this step runs the eager chain
            enc = eng.encode_first_stage

            def tagged_encode(x):      # the frozen VAE encoder's launches are reported apart from the UNet's ("nk_conv2d_fwd[frozen VAE]")
                timer.tag = "[frozen VAE]"
                try:
                    return enc(x)
                finally:
                    timer.tag = ""

            eng.encode_first_stage = tagged_encode
            try:
                step()
            finally:
                del eng.encode_first_stage      # (the instance attribute shadowing the method)
                est.wgrad_stream = side
                timer.uninstall()
                if graph_env is None:
                    os.environ.pop("NK_GRAPH", None)
                else:
                    os.environ["NK_GRAPH"] = graph_env
            return timer.summary() + (timer.overhead_ms,)

        f, ms, n, per, ovh = replay(serialized=bool(args.serialize))
        fs, mss, ns, pers, ovh_s = replay(serialized=True)
        ach, ach_s = f / (ms * 1e-3) / 1e12, fs / (mss * 1e-3) / 1e12
        traffic, traffic_src = pmc_traffic()
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "frac_in_step": round(ach / PEAK_BF16_TFLOPS, 4), "achieved_serialized": round(ach_s, 2), "frac_serialized": round(ach_s / PEAK_BF16_TFLOPS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": "nk_gemm_*_kernel<*> + nk_conv3x3_halo_kernel<*> (MFMA tile engine: linear+conv fwd/dgrad/wgrad)", "launches_per_step": n,
                    "avg_launch_us": round(ms * 1e3 / n, 2), "avg_launch_us_serialized": round(mss * 1e3 / ns, 2), "algorithmic_tflop_per_step": round(f / 1e12, 2),
                    "algorithmic_gflop_per_launch": round(f / n / 1e9, 2), "kernel_ms_per_step": round(ms, 2), "kernel_ms_per_step_serialized": round(mss, 2),
                    "by_entry_point": {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in per.items()},
                    "by_entry_point_serialized": {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in pers.items()},
                    "step_frac_of_mfma_peak": round(value / world * TFLOP_PER_IMAGE / PEAK_BF16_TFLOPS, 4),
                    "event_pair_overhead_us": round(0.5 * (ovh + ovh_s) * 1e3, 2),
                    "note": "launch intervals are HIP event pairs on the launch stream minus the measured empty-pair overhead. Serialized intervals are the "
                            "kernels' own durations (compare profiles/rNN_gemm_serialized_kernel_stats.csv). In-step intervals also contain the time a "
                            "launch waits for compute units held by the other stream's workgroups, which a kernel trace does not count as kernel "
                            "time: the trace's in-step average is shorter than avg_launch_us (DESIGN section 5)"}
    if world > 1:
        dist.barrier()

    steady_peak = torch.cuda.max_memory_allocated()
    emitted = []

    def emit(cpu):
        """rank 0's ONE JSON line (idempotent: the watchdog of the alternative-exchange phases may get here first)"""
        if emitted:
            return
        emitted.append(True)
        out = {
            "metric": "train images/sec (node) SDXL 1024^2", "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"SDXL-base {'mixed-res buckets (~1024^2 pixels)' if args.mixed_res else str(args.res) + '^2'} bf16, batch/GPU={args.batch}, UNet fwd+bwd + frozen VAE encode + {'Adafactor' if args.optimizer == 'adafactor' else 'AdamW'} step, {'frozen TE outputs synthetic' if args.precomputed_te else 'frozen CLIP-L + OpenCLIP-bigG conditioner on synthetic token ids'}",
                       "global_batch": args.batch * world * args.accumulate, "parallelism": f"dp{world}" + (" (exchange forced through RCCL at world 1: NK_DP_FORCE)" if forced else ""), "allreduce_dtype": args.wire_dtype, "activation_checkpointing": False if args.recompute == "none" else f"selective ({args.recompute})", "accumulate_grad_batches": args.accumulate},
            "loss": round(loss_val, 5), "max_mem_gb": round(max(steady_peak, setup_peak) / 2**30, 1), "max_mem_steady_gb": round(steady_peak / 2**30, 1),
            "max_mem_setup_gb": round(setup_peak / 2**30, 1),
            "step_ms_p50": round(pct(0.5), 2), "step_ms_p90": round(pct(0.9), 2), "step_ms_in_order": [round(t, 1) for t in in_order], "comm": comm,
            "host": {"unet_chain": "hipGraph replay" if priming else "eager launches", "graph_priming_steps": priming},
            "stream_k_fixup_timeouts": lib.query("nk_gemm_sk_status"),   # 0: every K-split tile was joined (gemm.hip)
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)

    alt_failed = False
    if comm is not None and args.alt_steps > 0:
        # an 8-GPU node is scarce, so the run that measures the default exchange also prices the other ones -- a few
        # steps each of {allreduce fp32, allreduce bf16 wire, rs_ag}, minus the configuration already timed.  rs_ag goes last (it leaves the
        # foreign parts' fp32 masters stale); a default of rs_ag makes its masters and statistics whole first.
        # These phases run collectives no multi-GPU box has executed yet (rs_ag on RCCL at world > 1), and the timed result above must survive
        # them: an exception is recorded in comm.alt and ends the phases (no further collective is attempted); a phase that does not finish
        # within --alt-timeout seconds makes EVERY rank's watchdog print (rank 0) and leave with exit code 0.
        import threading

        def bail():
            comm["alt_error"] = f"an alternative exchange configuration did not finish within {args.alt_timeout} s; the line was printed without it"
            if rank == 0:
                emit(None)
            sys.stdout.flush()
            os._exit(0)

        alts = [(m, w) for m, w in (("allreduce", "fp32"), ("allreduce", "bf16"), ("rs_ag", "fp32")) if (m, w) != (dp.mode, args.wire_dtype)]
        if args.optimizer != "adafactor":
            alts = [a for a in alts if a[0] != "rs_ag"]
        comm["alt"] = []
        for mode_, wire_ in alts:
            timer = threading.Timer(args.alt_timeout, bail)
            timer.daemon = True
            timer.start()
            try:
                if dp.sharded:
                    dp.sync_masters()
                    eng.adafactor.owned = None          # every rank runs the whole update again (chunks stay cut: no arithmetic changes)
                dp = FlatDataParallel(unet, eng.store, wire_dtype=torch.bfloat16 if wire_ == "bf16" else None, mode=mode_, broadcast_params=False)
                if dp.sharded:
                    dp.attach_optimizer(eng.adafactor)
                step()                                  # one untimed step: staging buffers, communicator warm-up of this collective
                dp.reducer.record_timing = True
                dp.reducer.take_counts()
                del comm_marks[:]
                barrier()
                ta = time.perf_counter()
                for _ in range(args.alt_steps):
                    step(mark=True)
                barrier()
                dta = time.perf_counter() - ta
                dp.reducer.record_timing = False
                tm = torch.tensor([dta], device=device, dtype=torch.float64)
                if world > 1:
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                entry = comm_summary(dp, comm_marks, wire_)
                entry.update({"steps": args.alt_steps, "ms_per_step": round(float(tm.item()) / args.alt_steps * 1e3, 2)})
                comm["alt"].append(entry)
            except Exception as e:  # noqa: BLE001 - the measured result must be reported whatever an untested collective path does
                comm["alt"].append({"mode": mode_, "wire_dtype": wire_, "error": f"{type(e).__name__}: {e}"[:400]})
                alt_failed = True
            finally:
                timer.cancel()
            if alt_failed:
                break
        if world > 1 and not alt_failed:
            dist.barrier()
    if (world > 1 or forced) and not alt_failed:
        # every collective is behind us: the group is taken down BEFORE rank 0 spends ~10 s on the CPU baseline, so that no rank sits in a
        # collective (or its watchdog) while another one is busy on the host
        dist.destroy_process_group()
    if rank == 0:
        emit(None if args.no_cpu_baseline or alt_failed else cpu_baseline())
    if alt_failed:      # other ranks may be parked in a collective this rank never joined: no orderly shutdown is possible
        sys.stdout.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
