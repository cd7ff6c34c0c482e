"""BASELINE configs[0] on the GPU: SD1.5 512x512, batch 1, one DiffusionEngine.training_step + optimizer step per iteration
(the configuration bench.py's cpu_baseline times on the host cores).  Prints ms/step and the speed-up over a CPU time given
on the command line (seconds), e.g.  python tools/bench_config1.py 4.6"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
import neurosis_amd.modules.diffusion as D
from neurosis_amd.models.autoencoder import AutoencoderKL
from neurosis_amd.models.diffusion import DiffusionEngine

dev = torch.device("cuda", 0)
torch.manual_seed(0)
with torch.device(dev):
    unet = D.UNetModel(**bench.SD15_UNET)
    vae = AutoencoderKL(embed_dim=4, ddconfig=bench.SDXL_VAE_DD)
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
den = den.to(dev)
bench.reinit_zero_modules(unet)
eng = DiffusionEngine(model=unet, denoiser=den, first_stage_model=vae, input_key="image", scale_factor=0.18215,
                      loss_fn=D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting()))
eng.setup_flat_params()
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
gen = torch.Generator(device=dev).manual_seed(1)
def step(B=1):
    batch = {"image": torch.rand(B, 3, 512, 512, device=dev, generator=gen) * 2 - 1, "crossattn": torch.randn(B, 77, 768, device=dev, generator=gen)}
    loss = eng.training_step(batch, 0, sigmas=torch.full((B,), 1.0, device=dev)); loss.backward(); eng.optimizer_step()
for B in (1, 16):
    for _ in range(3): step(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step(B)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    cpu = float(sys.argv[1]) if len(sys.argv) > 1 else None
    print(f"SD1.5 512^2 batch {B}: {ms:.1f} ms/step = {B / ms * 1e3:.1f} images/s, {3.53 * B / (ms * 1e-3):.0f} algorithmic TFLOP/s" + (f"; CPU oracle {cpu:.2f} s/step for batch 1 -> x{cpu * 1e3 / (ms / B):.0f} per image" if cpu else ""))
