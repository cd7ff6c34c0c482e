import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.lib import call
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, I in [(4096, 1280, 5120), (16384, 640, 2560)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, I, device="cuda") * I ** -0.5).to(torch.bfloat16)
    u = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16)
    du = torch.empty_like(u)
    t = bench(lambda: call("nk_linear_dgrad_geglu", dy.data_ptr(), w.data_ptr(), u.data_ptr(), du.data_ptr(), M, N, I, N, I, 2 * I, 2 * I, ops._stream()))
    print(f"dgrad_geglu M={M} N={N} I={I}: {t*1e6:.1f} us = {2.0*M*N*I/t/1e12:.0f} TFLOP/s")
