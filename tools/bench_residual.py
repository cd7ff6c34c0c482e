import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
def timed(fns, iters):
    for f in fns[:8]: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, K in [(4096, 1280, 1280), (4096, 1280, 5120), (16384, 640, 640), (4096, 3840, 1280)]:
    R = 48
    xs, ws, rs = [rb(M, K) for _ in range(R)], [rb(N, K) for _ in range(R)], [rb(M, N) for _ in range(R)]
    dys, adds = [rb(M, N) for _ in range(R)], [rb(M, K) for _ in range(R)]
    b = torch.randn(N, device="cuda")
    f = min(timed([(lambda j=j: ops.gemm_nt(xs[j], ws[j], b, rs[j])) for j in range(R)], 192) for _ in range(3))
    d = min(timed([(lambda j=j: ops.gemm_nn(dys[j], ws[j], adds[j])) for j in range(R)], 192) for _ in range(3))
    print(f"{M} x {N} x {K}: fwd + bias + residual {f:6.1f} us | dgrad + dx_add {d:6.1f} us", flush=True)
