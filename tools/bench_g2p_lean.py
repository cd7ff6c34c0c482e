"""A/B of the lean producer sources of the two-group kernel (NK_GEMM_LEAN=1, round 6) against OpG2::next_sources in the producer waves (=0) on the shapes
that kernel takes (NK_GEMM_G2=2: every eligible launch), interleaved rounds in one process, serialized launches, random data; results compared too."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
os.environ["NK_GEMM_G2"] = "2"
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
def time_modes(fn, iters=30, rounds=3):
    res = {"0": [], "1": []}
    for _ in range(rounds):
        for mode in ("0", "1"):
            os.environ["NK_GEMM_LEAN"] = mode
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters): fn()
            e.record(); torch.cuda.synchronize()
            res[mode].append(s.elapsed_time(e) / iters * 1e3)
    return min(res["0"]), min(res["1"])
LIN = [(4096, 1280, 1280), (4096, 3840, 1280), (4096, 1280, 5120), (16384, 640, 640), (16384, 1920, 640), (16384, 640, 2560), (4096, 2560, 1280), (4096, 10240, 1280), (1000, 1280, 1280)]
print(f"{'kind':6s} {'M':>6s} {'N':>6s} {'K':>6s}   {'old us':>8s} {'lean us':>8s}   {'old TF':>7s} {'leanTF':>7s}  ratio  maxdiff")
for M, N, K in LIN:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    for kind, fn in (("fwd", lambda: ops.gemm_nt(x, w)), ("dgrad", lambda: ops.gemm_nn(dy, w)), ("wgrad", lambda: (ops.gemm_tn_f32(dy, x, dw, False), dw)[1])):
        os.environ["NK_GEMM_LEAN"] = "0"; a = fn().float().clone()
        os.environ["NK_GEMM_LEAN"] = "1"; b = fn().float().clone()
        t0, t1 = time_modes(fn)
        print(f"{kind:6s} {M:6d} {N:6d} {K:6d}   {t0:8.1f} {t1:8.1f}   {fl/t0/1e6:7.0f} {fl/t1/1e6:7.0f}  {t0/t1:5.2f}  {(a - b).abs().max().item():.3g}", flush=True)
