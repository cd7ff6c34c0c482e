"""us per launch of the text encoders' GEMMs (308 = 4 x 77 token rows) under the stream-K policy switch."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
def t(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
shapes = [(308, 3840, 1280), (308, 1280, 1280), (308, 5120, 1280), (308, 1280, 5120), (308, 2304, 768), (308, 768, 768), (308, 3072, 768), (308, 768, 3072), (308, 2560, 2048)]
modes = sys.argv[1:] or ["4", "1", "0"]
print("M x N x K".ljust(20) + "".join(f"  SK={m:>2s} us" for m in modes))
for M, N, K in shapes:
    x, w = rb(M, K), rb(N, K)
    b = torch.zeros(N, device="cuda")
    row = f"{M} x {N} x {K}".ljust(20)
    for m in modes:
        os.environ["NK_GEMM_SK"] = m
        row += f"  {t(lambda: ops.gemm_nt(x, w, b)):9.1f}"
    os.environ.pop("NK_GEMM_SK", None)
    print(row + f"    ({2.0 * M * N * K / 1e9:.1f} GFLOP)", flush=True)
