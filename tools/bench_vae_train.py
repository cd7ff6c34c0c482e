"""BASELINE config 5's reconstruction part on one GPU: the SD/SDXL autoencoder (ch 128, mult 1-2-4-4, 84 M parameters) trained at
256x256 -- encode, sample the posterior, decode, l2 loss, backward, fused AdamW -- through AutoencodingEngine.training_step.
With --gan the PatchGAN discriminator (ndf 64, 3 layers, 2.8 M parameters) joins in: autoencoder step (nll + adaptive-weight
adversarial term) and discriminator step alternate, one image batch each; --lpips adds the LPIPS term (AlexNet trunk, the reference default; --lpips-vgg for VGG16) to the autoencoder
step (random trunk weights here: the ImageNet weights are a download).   python tools/bench_vae_train.py [--gan] [--lpips] [batch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from neurosis_amd.models.autoencoder import AutoencodingEngine, DiagonalGaussianRegularizer
from neurosis_amd.modules.diffusion.model import Decoder, Encoder

from neurosis_amd.modules.losses import NLayerDiscriminator

gan, lpips = "--gan" in sys.argv, "--lpips" in sys.argv or "--lpips-vgg" in sys.argv
trunk = "vgg" if "--lpips-vgg" in sys.argv else "alex"
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
dd = dict(bench.SDXL_VAE_DD, standalone=True)
with torch.device(dev):
    disc = NLayerDiscriminator().initialize_weights() if gan else None
    from neurosis_amd.modules.losses import LPIPS
    perceptual = LPIPS(pnet_type=trunk, pretrained=False, pnet_rand=True) if lpips else None
    eng = AutoencodingEngine(encoder=Encoder(**dd, embed_dim=4), decoder=Decoder(**dd, embed_dim=4), loss="l2", regularizer=DiagonalGaussianRegularizer(),
                             discriminator=disc, perceptual_loss=perceptual, regularization_weights={"kl_loss": 1e-6})
eng = eng.to(dev)
eng.setup_flat_params()
counter = [0]
params = sum(p.numel() for p in eng.parameters())
gen = torch.Generator(device=dev).manual_seed(1)
for B in [int(a) for a in argv] or [8, 32]:
    def step():
        counter[0] += 1
        return eng.training_step({"image": torch.rand(B, 3, 256, 256, device=dev, generator=gen) * 2 - 1}, counter[0])
    for _ in range(3): loss = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): loss = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    timer = bench.GemmTimer(); timer.install()          # algorithmic FLOPs of the step = sum of 2*M*N*K over its tile-engine launches
    try:
        step(); torch.cuda.synchronize()
    finally:
        timer.uninstall()
    tflop = sum(r[1] for r in timer.records) / 1e12
    print(f"VAE 256^2 {'with PatchGAN (alternating steps) ' if gan else ''}{'+ LPIPS(' + trunk + ') ' if lpips else ''}batch {B}: {ms:.1f} ms/step = {B / ms * 1e3:.0f} images/s, {tflop:.1f} algorithmic TFLOP per step in the tile engine = {tflop / ms * 1e3:.0f} TFLOP/s; loss {float(loss):.4f}; "
          f"{params / 1e6:.1f} M parameters, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
