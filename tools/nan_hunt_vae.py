"""tools/nan_hunt.py for BASELINE config 5: AutoencodingEngine.training_step at 256 x 256 with the PatchGAN discriminator and LPIPS (autoencoder and
discriminator steps alternating, fused AdamW), run for many steps; after every step the loss and every parameter / gradient buffer are checked
for finiteness, and the loss for a jump of more than 50 x its running median.   python tools/nan_hunt_vae.py [steps] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from neurosis_amd.models.autoencoder import AutoencodingEngine, DiagonalGaussianRegularizer  # noqa: E402
from neurosis_amd.modules.diffusion.model import Decoder, Encoder  # noqa: E402
from neurosis_amd.modules.losses import LPIPS, NLayerDiscriminator  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
torch.manual_seed(0)
dd = dict(bench.SDXL_VAE_DD, standalone=True)
with torch.device(dev):
    disc = NLayerDiscriminator().initialize_weights()
    perceptual = LPIPS(pnet_type="alex", pretrained=False, pnet_rand=True)
    eng = AutoencodingEngine(encoder=Encoder(**dd, embed_dim=4), decoder=Decoder(**dd, embed_dim=4), loss="l2", regularizer=DiagonalGaussianRegularizer(),
                             discriminator=disc, perceptual_loss=perceptual, regularization_weights={"kl_loss": 1e-6})
eng = eng.to(dev)
eng.setup_flat_params()
gen = torch.Generator(device=dev).manual_seed(1)
hist = {0: [], 1: []}      # autoencoder steps and discriminator steps alternate: their losses are judged apart
for s in range(1, STEPS + 1):
    loss = eng.training_step({"image": torch.rand(B, 3, 256, 256, device=dev, generator=gen) * 2 - 1}, s)
    v = float(loss.detach())
    bad = [n for n, p in eng.named_parameters() if not bool(torch.isfinite(p).all())]
    gbad = [n for n, p in eng.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    h = hist[s & 1]
    med = sorted(h[-40:])[len(h[-40:]) // 2] if len(h) >= 10 else None
    if v != v or abs(v) == float("inf") or bad or gbad or (med is not None and abs(v) > 50 * max(abs(med), 1e-3) and not os.environ.get("NK_HUNT_VERBOSE")):
        print(f"step {s}: loss {v}  (running median {med});  parameters with non-finite values {len(bad)} {bad[:4]};  gradients {len(gbad)} {gbad[:4]}", flush=True)
        for n, p in eng.named_parameters():
            if p.grad is not None and p.grad.ndim >= 2:
                g = p.grad.float()
                if not bool(torch.isfinite(g).all()) or float(g.abs().max()) > 1e4 * max(float(g.abs().mean()), 1e-12) and float(g.abs().max()) > 1e6:
                    print(f"   {n} {tuple(g.shape)}: max |g| {float(g.abs().max()):.4g}  mean |g| {float(g.abs().mean()):.4g}", flush=True)
        break
    h.append(v)
    if s <= 30 and os.environ.get("NK_HUNT_VERBOSE"):
        log = getattr(eng, "last_log", None) or {}
        print(f"   step {s}: loss {v:.5g}  " + "  ".join(f"{k.split('/')[-1]} {float(x):.4g}" for k, x in log.items() if torch.is_tensor(x) and x.numel() == 1), flush=True)
    if s % 200 == 0:
        print(f"   ... step {s}  loss {v:.4f}", flush=True)
else:
    print(f"{STEPS} steps (autoencoder / discriminator alternating, batch {B}): all finite, no loss jump; last losses {hist[0][-1]:.4f} / {hist[1][-1]:.4f}", flush=True)
