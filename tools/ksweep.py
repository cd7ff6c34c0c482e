"""Fixed per-tile cost vs per-k-step cost of the tile engine: time(K) at a fixed tile grid, linear fit."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
import numpy as np

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
for M, N in [(16384, 5120), (16384, 640), (4096, 1280), (8192, 1024)]:
    tiles = (M // 128) * ((N + 127) // 128)
    Ks = [128, 256, 512, 1024, 2048, 4096]
    for mode in ("fwd", "dgrad"):
        ts = []
        for K in Ks:
            x, w = rb(M, K), rb(N, K)
            if mode == "fwd":
                ts.append(timeit(lambda: ops.gemm_nt(x, w)))
            else:
                dy = rb(M, N)      # dx[M,K] = dy[M,N] @ w[N,K]: reduction over N -> sweep the reduction by swapping roles
                w2 = rb(K, N)      # reduction length K, output [M, N]
                dyk = rb(M, K)
                ts.append(timeit(lambda: ops.gemm_nn(dyk, w2)))
        b, a = np.polyfit(np.array(Ks) / 64.0, np.array(ts), 1)
        waves = -(-tiles // 512)
        print(f"M={M} N={N} tiles={tiles} ({waves} rounds of 512) {mode}: " + " ".join(f"K{K}={t:.1f}" for K, t in zip(Ks, ts)) +
              f" | fit: fixed {a:.1f} us + {b:.2f} us/kstep  -> per round: fixed {a/waves:.2f} us, kstep {b/waves*1e3:.0f} ns; ideal kstep (2 blocks/CU, 100% MFMA @2.0GHz) = 512 ns")
