"""Head-dim-512 attention backward: the flash kernels of csrc/attn512_bwd.h against the chunked recompute through HBM (NK_ATTN512_BWD=0).
usage (GPU box): python tools/bench_attn512_bwd.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for B, L in [(4, 1024), (4, 4096), (1, 16384)]:
    D = 512
    q, k, v, do = (torch.randn(B * L, D, device="cuda").to(torch.bfloat16) for _ in range(4))
    fl = 10.0 * B * L * L * D
    res = []
    for impl in ("1", "0"):
        os.environ["NK_ATTN512_BWD"] = impl
        o, bwd = ops.attention512_fwd(q, k, v, B)
        t = timeit(lambda: bwd(do), 3 if L > 4096 else 5)
        res.append(t)
    print(f"B={B} L={L}: flash backward {res[0]:10.1f} us ({fl / res[0] / 1e6:6.0f} TFLOP/s algorithmic) | chunked recompute {res[1]:10.1f} us ({fl / res[1] / 1e6:6.0f})")
