"""The frozen text towers' GEMMs (308 rows = 4 prompts x 77 tokens): weight-streaming launches of a few dozen tiles.  us per launch with the
weights rotating over 48 buffers (a tower reads every weight once): the 64 x 64 eight-stage ring kernel (default for such launches) against
NK_GEMM_R64=0 under the stream-K policies (NK_GEMM_SK: 4 = by shape; 1 = always; 0 = never).  Launches are replayed from a hipGraph: launched one
by one from Python they are host-bound (~11 us each)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape, s=0.5): return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
def timed(fns, iters):
    for f in fns[:8]: f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        for f in fns[:2]: f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for i in range(iters): fns[i % len(fns)]()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = 308
for N, K in [(3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    R = 48
    x, ws, b, r = rb(M, K), [rb(N, K, s=K ** -0.5) for _ in range(R)], torch.randn(N, device="cuda"), rb(M, N)
    row = []
    os.environ["NK_GEMM_R64"] = "1"
    row.append(min(timed([(lambda w=w: ops.gemm_nt(x, w, b, r)) for w in ws], 96) for _ in range(2)))
    ref = ops.gemm_nt(x, ws[0], b, r).float()
    os.environ["NK_GEMM_R64"] = "0"
    for sk in ("4", "1", "0"):
        os.environ["NK_GEMM_SK"] = sk
        row.append(min(timed([(lambda w=w: ops.gemm_nt(x, w, b, r)) for w in ws], 96) for _ in range(2)))
    os.environ.pop("NK_GEMM_SK")
    diff = float((ops.gemm_nt(x, ws[0], b, r).float() - ref).abs().max())
    os.environ.pop("NK_GEMM_R64")
    print(f"{M} x {N} x {K}: 64 x 64 ring {row[0]:6.1f} us | 128 x 128 kernels: by shape {row[1]:6.1f} | always stream-K {row[2]:6.1f} | never {row[3]:6.1f}   "
          f"({N * K * 2 / 1e6:.1f} MB of weights: {N * K * 2 / row[0] / 1e3:.0f} GB/s; max diff {diff:.3g})", flush=True)
