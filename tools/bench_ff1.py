"""FeedForward projection with the GEGLU epilogue (nk_linear_fwd_geglu, 256 x 256 two-group kernel) at the two SDXL widths, weights rotating over 24 buffers."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape, s=0.5): return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
for M, I, K in [(4096, 5120, 1280), (16384, 2560, 640)]:
    R = 24
    x = rb(M, K)
    ws = [torch.nn.Parameter(torch.randn(2 * I, K, device="cuda") * K ** -0.5, requires_grad=False) for _ in range(R)]
    b = torch.nn.Parameter(torch.randn(2 * I, device="cuda") * 0.1, requires_grad=False)
    for w in ws: ops.linear_geglu_fwd(x, w, b)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(96): ops.linear_geglu_fwd(x, ws[i % R], b)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 96 * 1e3)
    print(f"linear_fwd_geglu {M} x {2 * I} x {K}: {best:6.1f} us = {2.0 * M * 2 * I * K / best / 1e6:5.0f} TFLOP/s", flush=True)
