"""The fused Adafactor update ALONE on the chip (nothing beside it), against its stream time in the step (beside the next step's frozen VAE
encoder): stream time, launches and effective TB/s over the bytes it touches (22 B/param: g read three times, p read + written, bf16 shadow
written; 14 B/param if the chunk's second and third reads of g come out of the Infinity Cache).   usage (GPU box): python tools/bench_optimizer.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench

dev = torch.device("cuda", 0)
eng = bench.build_engine(dev, (1024, 1024), None)
eng.configure_adafactor(scale_parameter=True, relative_step=True, warmup_init=True)
eng.overlap_optimizer = False                      # in line on the current stream: what is timed is the update itself
eng.store.grad.normal_(0, 1e-3)
n = eng.store.grad.numel()
for chunk_mb in (os.environ.get("NK_AF_CHUNK_MB", "default"),):
    for _ in range(2):
        eng.optimizer_step(lr=1e-6, weight_decay=1e-2, grad_scale=1.0, dp=None)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        eng.optimizer_step(lr=1e-6, weight_decay=1e-2, grad_scale=1.0, dp=None)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"Adafactor alone, chunk {chunk_mb} MB: {ms:.2f} ms per update of {n / 1e9:.3f} G parameters = {22 * n / ms / 1e9:.2f} TB/s over 22 B/param touched, "
          f"{14 * n / ms / 1e9:.2f} TB/s over 14 B/param (HBM floor at 8 TB/s: {14 * n / 8e9:.2f} ms)")
