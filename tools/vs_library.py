"""The tile engine against the vendor library on the step's main Linear shapes (context, not a dependency: the product never
calls it): torch.matmul (hipBLASLt / rocBLAS underneath) vs ops.gemm_nt, bf16 in, fp32 accumulate, bf16 out, random data."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

shapes = [(16384, 640, 640), (16384, 5120, 640), (16384, 640, 2560), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120),
          (65536, 1280, 1280), (308, 1280, 1280)]
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lib = bench(lambda: torch.matmul(x, w.t(), out=out))
    t_mine = bench(lambda: ops.gemm_nt(x, w, out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:5d}: library {t_lib * 1e6:7.1f} us = {fl / t_lib / 1e12:6.0f} TFLOP/s | tile engine {t_mine * 1e6:7.1f} us = {fl / t_mine / 1e12:6.0f} TFLOP/s"
          f" | ratio {t_lib / t_mine:.2f}", flush=True)

if "--wgrad" in sys.argv or "--all" in sys.argv:
    # weight gradients dW[N, K] = dy[M, N]^T x[M, K] (fp32 out here, bf16 out from the library: an upper bound for it) and
    # input gradients dx[M, K] = dy[M, N] w[N, K]
    for M, N, K in [(4096, 1280, 1280), (4096, 3840, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (16384, 640, 640), (16384, 5120, 640), (16384, 640, 2560)]:
        dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * N ** -0.5).to(torch.bfloat16)
        dw = torch.zeros(N, K, device="cuda")
        dwl = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
        dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        t_lib = bench(lambda: torch.matmul(dy.t(), x, out=dwl))
        t_mine = bench(lambda: ops.gemm_tn_f32(dy, x, dw, False))
        t_libd = bench(lambda: torch.matmul(dy, w, out=dx))
        t_mined = bench(lambda: ops.gemm_nn(dy, w, out=dx))
        print(f"M={M:6d} N={N:6d} K={K:5d}: wgrad library {fl / t_lib / 1e12:6.0f} | tile engine {fl / t_mine / 1e12:6.0f} TFLOP/s ({t_mine * 1e6:6.1f} us)"
              f"   dgrad library {fl / t_libd / 1e12:6.0f} | tile engine {fl / t_mined / 1e12:6.0f} TFLOP/s ({t_mined * 1e6:6.1f} us)", flush=True)
