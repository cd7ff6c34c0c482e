"""BASELINE configs[0] (SD1.5 512^2, batch 1) for a kernel trace: N steps of training_step + optimizer (see bench_config1.py)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
def step():
    batch = {"image": torch.rand(B, 3, 512, 512, device=dev, generator=gen) * 2 - 1, "crossattn": torch.randn(B, 77, 768, device=dev, generator=gen)}
    loss = eng.training_step(batch, 0, sigmas=torch.full((B,), 1.0, device=dev)); loss.backward(); eng.optimizer_step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / N * 1e3:.1f} ms/step")
