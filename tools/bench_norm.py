"""HBM-bound kernels at the SDXL 1024^2 / batch-4 shapes: achieved GB/s against algorithmic bytes (read once, write once)."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rb(*shape): return (torch.randn(*shape, device="cuda")).to(torch.bfloat16)
print(f"{'kernel':34s} {'shape':22s} {'us':>9s} {'GB/s':>8s}  (algorithmic bytes)")
for (N, H, W, C) in [(4, 128, 128, 320), (4, 128, 128, 960), (4, 64, 64, 640), (4, 64, 64, 1920), (4, 32, 32, 1280), (4, 32, 32, 2560), (4, 1024, 1024, 128), (4, 512, 512, 256)]:
    x = ops.Img(rb(N * H * W, C), N, H, W)
    g = torch.nn.Parameter(torch.ones(C, device="cuda")); b = torch.nn.Parameter(torch.zeros(C, device="cuda"))
    n = N * H * W * C
    t = timeit(lambda: ops.groupnorm_fwd(x, g, b, 32, 1e-5, True), 10)
    print(f"{'groupnorm+silu fwd':34s} {str((N,C,H,W)):22s} {t*1e3:9.1f} {4*n/t/1e6:8.0f}  (2 B read + 2 B write per element)")
    if H <= 128:
        out, bwd = ops.groupnorm_fwd(x, g, b, 32, 1e-5, True)
        dy = rb(N * H * W, C)
        t = timeit(lambda: bwd(dy), 10)
        print(f"{'groupnorm+silu bwd':34s} {str((N,C,H,W)):22s} {t*1e3:9.1f} {6*n/t/1e6:8.0f}  (4 B read + 2 B write per element)")
for (M, C) in [(16384, 640), (4096, 1280)]:
    x = rb(M, C); g = torch.nn.Parameter(torch.ones(C, device="cuda")); b = torch.nn.Parameter(torch.zeros(C, device="cuda"))
    n = M * C
    t = timeit(lambda: ops.layernorm_fwd(x, g, b))
    print(f"{'layernorm fwd':34s} {str((M,C)):22s} {t*1e3:9.1f} {4*n/t/1e6:8.0f}")
    out, bwd = ops.layernorm_fwd(x, g, b); dy = rb(M, C)
    for fused in ("1", "0"):      # one-pass rows kernel + partial-row reduce  vs  dx | parameter pass | column reduce
        os.environ["NK_LN_FUSED"] = fused
        t = timeit(lambda: bwd(dy, dy))
        print(f"{'layernorm bwd (+residual grad) ' + ('one pass' if fused == '1' else '3 kernels'):34s} {str((M,C)):22s} {t*1e3:9.1f} {8*n/t/1e6:8.0f}")
    os.environ.pop("NK_LN_FUSED")
for (M, I) in [(16384, 2560), (4096, 5120)]:
    u = rb(M, 2 * I); n = M * I
    t = timeit(lambda: ops.geglu_fwd(u))
    print(f"{'geglu fwd':34s} {str((M,2*I)):22s} {t*1e3:9.1f} {6*n/t/1e6:8.0f}")
    y, bwd = ops.geglu_fwd(u); dy = rb(M, I)
    t = timeit(lambda: bwd(dy))
    print(f"{'geglu bwd':34s} {str((M,2*I)):22s} {t*1e3:9.1f} {10*n/t/1e6:8.0f}")
# attention at the SDXL shapes: TFLOP/s
for (B, Hh, Lq, Lk) in [(4, 10, 4096, 4096), (4, 20, 1024, 1024), (4, 10, 4096, 77), (4, 20, 1024, 77)]:
    D = 64
    q, k, v = rb(B * Lq, Hh * D), rb(B * Lk, Hh * D), rb(B * Lk, Hh * D)
    fl = 4.0 * B * Hh * Lq * Lk * D
    t = timeit(lambda: ops.attention_fwd(q, k, v, B, Hh, D), 10)
    o, bwd = ops.attention_fwd(q, k, v, B, Hh, D); do = rb(B * Lq, Hh * D)
    tb = timeit(lambda: bwd(do), 10)
    print(f"{'attention fwd / bwd':34s} {str((B,Hh,Lq,Lk)):22s} {t*1e3:9.1f} us {fl/t/1e9:7.0f} TF/s | bwd {tb*1e3:9.1f} us {2.5*fl/tb/1e9:7.0f} TF/s (algorithmic 2.5x fwd)")
