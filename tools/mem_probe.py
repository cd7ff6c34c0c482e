"""Where the step's peak memory is: allocated bytes at the end of the UNet forward (eager, NK_GRAPH=0) under each recompute policy.
Run on the GPU box: NK_GRAPH=0 python tools/mem_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NK_GRAPH", "0")
import torch

import bench
from neurosis_amd import ops

dev = torch.device("cuda:0")
for policy in (None, "norms"):
    torch.manual_seed(0)
    eng = bench.build_engine(dev, (1024, 1024), None)
    unet = eng.model.diffusion_model
    unet.set_recompute(policy)
    gen = torch.Generator(device=dev).manual_seed(0)
    batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen, True)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    marks = {}
    orig = unet.fwd

    def fwd(*a, **k):
        out = orig(*a, **k)
        torch.cuda.synchronize()
        marks["end_fwd"] = torch.cuda.memory_allocated()
        return out

    unet.fwd = fwd
    torch.cuda.reset_peak_memory_stats()
    loss = eng.training_step(batch, 0, sigmas=bench.draw_sigmas(4, gen, dev))
    loss.backward()
    torch.cuda.synchronize()
    print(f"policy={policy}: resident before step {base / 2**30:.2f} GiB, end of UNet forward {marks.get('end_fwd', 0) / 2**30:.2f} GiB, "
          f"peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, loss {float(loss):.6f}", flush=True)
    del eng, unet, batch, loss, orig, fwd
    import gc

    gc.collect()
    torch.cuda.empty_cache()
