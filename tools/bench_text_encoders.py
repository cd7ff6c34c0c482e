"""SDXL's two frozen text encoders at full size on one GPU: CLIP ViT-L/14 (12 layers, width 768) and OpenCLIP ViT-bigG/14 (32 layers,
width 1280) on 77-token prompts, through GeneralConditioner with the size/crop embedders, random weights, token ids as input.
python tools/bench_text_encoders.py [batch ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2
from neurosis_amd.modules.encoders import ConcatTimestepEmbedderND, GeneralConditioner

dev = torch.device("cuda", 0)
torch.manual_seed(0)
clip_l = FrozenCLIPEmbedder(layer="hidden", layer_idx=11, input_key="ids")
big_g = FrozenOpenCLIPEmbedder2(arch="ViT-bigG-14", layer="penultimate", always_return_pooled=True, input_key="ids")
sizes = [ConcatTimestepEmbedderND(outdim=256, input_key=k) for k in ("original_size_as_tuple", "crop_coords_top_left", "target_size_as_tuple")]
cond = GeneralConditioner([clip_l, big_g, *sizes]).to(dev)
params = sum(p.numel() for p in cond.parameters())
for B in [int(a) for a in sys.argv[1:]] or [1, 4, 16]:
    ids = torch.randint(3, 49000, (B, 77), device=dev); ids[:, 0] = 49406; ids[:, 20:] = 49407
    batch = {"ids": ids, "original_size_as_tuple": torch.full((B, 2), 1024.0, device=dev), "crop_coords_top_left": torch.zeros(B, 2, device=dev),
             "target_size_as_tuple": torch.full((B, 2), 1024.0, device=dev)}
    out = cond(batch); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): out = cond(batch)
    host = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    flops = 2 * B * 77 * (12 * (12 * 768**2) + 32 * (12 * 1280**2))      # projections + MLP; attention itself is negligible at L = 77
    print(f"batch {B}: conditioner {ms:.2f} ms (host enqueue {host * 1e3:.2f} ms), {flops / ms / 1e9:.1f} TFLOP/s; crossattn {tuple(out['crossattn'].shape)} "
          f"vector {tuple(out['vector'].shape)} finite={bool(torch.isfinite(out['crossattn']).all())}; {params / 1e6:.0f} M parameters", flush=True)
