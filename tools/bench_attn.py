"""Attention microbench: forward / backward TFLOP/s at chosen (B, H, Lq, Lk), D = 64."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rb(*shape): return torch.randn(*shape, device="cuda").to(torch.bfloat16)
shapes = [(4, 20, 1024, 77), (4, 10, 4096, 77), (4, 20, 1024, 1024), (8, 20, 1024, 1024), (16, 20, 1024, 1024), (4, 10, 4096, 4096), (2, 10, 4096, 4096), (1, 10, 4096, 4096)]
for (B, Hh, Lq, Lk) in shapes:
    D = 64
    q, k, v = rb(B * Lq, Hh * D), rb(B * Lk, Hh * D), rb(B * Lk, Hh * D)
    fl = 4.0 * B * Hh * Lq * Lk * D
    t = timeit(lambda: ops.attention_fwd(q, k, v, B, Hh, D))
    o, bwd = ops.attention_fwd(q, k, v, B, Hh, D); do = rb(B * Lq, Hh * D)
    tb = timeit(lambda: bwd(do))
    print(f"B={B:2d} H={Hh} Lq={Lq} Lk={Lk}: fwd {t*1e3:7.1f} us {fl/t/1e9:6.0f} TF/s | bwd {tb*1e3:7.1f} us {2.5*fl/tb/1e9:6.0f} TF/s (algorithmic 2.5x)  blocks fwd={(Lq//128)*Hh*B}")
