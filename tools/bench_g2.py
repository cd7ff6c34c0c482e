"""A/B of the two-group staggered kernel (NK_GEMM_G2=2) against the kernels it replaces (=0) on the SDXL Linear shapes:
interleaved rounds in one process, serialized launches, random data.  Prints us per launch and TFLOP/s for both."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)

def time_modes(fn, iters=30, rounds=3):
    res = {"0": [], "2": []}
    for _ in range(rounds):
        for mode in ("0", "2"):
            os.environ["NK_GEMM_G2"] = mode
            for _ in range(3): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters): fn()
            e.record(); torch.cuda.synchronize()
            res[mode].append(s.elapsed_time(e) / iters * 1e3)
    return min(res["0"]), min(res["2"])

LIN = [(4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (16384, 640, 640), (16384, 1920, 640), (16384, 5120, 640),
       (16384, 640, 2560), (4096, 2560, 1280), (16384, 1280, 640), (65536, 1280, 1280), (4096, 4096, 4096)]
print(f"{'kind':6s} {'M':>6s} {'N':>6s} {'K':>6s}   {'old us':>8s} {'g2 us':>8s}   {'old TF':>7s} {'g2 TF':>7s}  ratio")
for M, N, K in LIN:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    for kind, fn in (("fwd", lambda: ops.gemm_nt(x, w)), ("dgrad", lambda: ops.gemm_nn(dy, w)), ("wgrad", lambda: ops.gemm_tn_f32(dy, x, dw, False))):
        t0, t2 = time_modes(fn)
        print(f"{kind:6s} {M:6d} {N:6d} {K:6d}   {t0:8.1f} {t2:8.1f}   {fl/t0/1e6:7.0f} {fl/t2/1e6:7.0f}  {t0/t2:5.2f}", flush=True)
# three 1280^2 weight gradients as one launch
import ctypes as C
M, N, K, n = 4096, 1280, 1280, 3
dys, xs, dws = [rb(M, N) for _ in range(n)], [rb(M, K) for _ in range(n)], [torch.zeros(N, K, device="cuda") for _ in range(n)]
arr = C.c_void_p * n
def batched():
    ops.call("nk_linear_wgrad_batched", arr(*[t.data_ptr() for t in dys]), arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dws]), None, n, M, N, K, N, K, K, 0, ops._stream())
t0, t2 = time_modes(batched)
fl = 2.0 * M * N * K * n
print(f"{'wgradx3':6s} {M:6d} {N:6d} {K:6d}   {t0:8.1f} {t2:8.1f}   {fl/t0/1e6:7.0f} {fl/t2/1e6:7.0f}  {t0/t2:5.2f}")
