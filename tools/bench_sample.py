"""SDXL-base 1024x1024 sampling on one GPU: EulerEDMSampler + VanillaCFG(7.5) as configs/sdxl/sdxl.example.yaml
names it, through DiffusionEngine.sample (FusedDenoiser -> nk_sample_* kernels) and, for comparison, through the generic route
(the reference's op-for-op arithmetic in torch ops around the same HIP UNet); then DiffusionEngine.decode_first_stage.
Random-init weights, synthetic conditioning.   python tools/bench_sample.py [steps] [batch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import neurosis_amd.modules.diffusion as D
import neurosis_amd.modules.diffusion.sampling as S
from neurosis_amd.models.autoencoder import AutoencoderKL
from neurosis_amd.models.diffusion import DiffusionEngine
from neurosis_amd.modules.guidance import VanillaCFG

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batches = [int(a) for a in sys.argv[2:]] or [1, 4]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    unet = D.UNetModel(**bench.SDXL_UNET)
    vae = AutoencoderKL(embed_dim=4, ddconfig=bench.SDXL_VAE_DD)
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
den = den.to(dev)
bench.reinit_zero_modules(unet)
sampler = S.EulerEDMSampler(discretization=D.LegacyDDPMDiscretization(), guider=VanillaCFG(7.5), num_steps=steps)
eng = DiffusionEngine(model=unet, denoiser=den, first_stage_model=vae, sampler=sampler, input_key="image", scale_factor=0.13025).eval()
for p in eng.parameters():
    p.requires_grad_(False)
gen = torch.Generator(device=dev).manual_seed(1)
UNET_FWD_TFLOP = 91.09 / 4 / 3          # bench.py's algorithmic count: fwd+bwd for batch 4 = 91.09 TFLOP, backward = 2x forward

host_s = 0.0
def timed(fn, n=2):
    """(seconds per call, last result); also leaves the host-side enqueue time per call in `host_s`"""
    global host_s
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    host_s = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, out

for B in batches:
    cond = {"crossattn": torch.randn(B, 77, 2048, device=dev, generator=gen), "vector": torch.randn(B, 2816, device=dev, generator=gen)}
    uc = {"crossattn": torch.zeros(B, 77, 2048, device=dev), "vector": torch.randn(B, 2816, device=dev, generator=gen)}
    noise = torch.randn(B, 4, 128, 128, device=dev, generator=gen)
    t_fused, x = timed(lambda: eng.sample(cond, uc=uc, batch_size=B, shape=(4, 128, 128), noise=noise))
    host_fused = host_s
    def generic():
        with torch.no_grad():
            return sampler(lambda i, s, c: eng.denoiser(eng.model, i, s, c, "D"), noise.clone(), cond, uc=uc)
    t_gen, xg = timed(generic)
    t_dec, img = timed(lambda: eng.decode_first_stage(x))
    diff = float((x - xg).abs().max() / xg.abs().max())
    print(f"SDXL 1024^2 batch {B} (CFG -> UNet batch {2 * B}), {steps} Euler steps: fused {t_fused / steps * 1e3:.1f} ms/step = {steps / t_fused:.2f} it/s "
          f"({2 * B * UNET_FWD_TFLOP / (t_fused / steps):.0f} TFLOP/s, host enqueue {host_fused / steps * 1e3:.1f} ms/step); generic route {t_gen / steps * 1e3:.1f} ms/step; routes differ by {diff:.1e}; "
          f"VAE decode {t_dec * 1e3:.0f} ms ({B / t_dec:.2f} images/s); end to end {B / (t_fused + t_dec):.3f} images/s; "
          f"finite={bool(torch.isfinite(img).all())} peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
