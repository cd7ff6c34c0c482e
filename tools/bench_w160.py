"""A/B of the exact-round 160-row weight-gradient kernel (csrc/gemm_w160.h, NK_GEMM_W160=1) against the kernels it replaces (=0) on the SDXL
Linear weight-gradient shapes: interleaved rounds in one process, serialized launches, random data, operands rotating over four buffer sets
(a training step never finds its dy / x in the Infinity Cache); the two results are compared as well.
    python tools/bench_w160.py            # run on the GPU box
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops  # noqa: E402

NSETS = 4


def rb(*shape):
    return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)


def time_modes(fn, iters=24, rounds=3):
    res = {"0": [], "1": []}
    for _ in range(rounds):
        for mode in ("0", "1"):
            os.environ["NK_GEMM_W160"] = mode
            for i in range(4):
                fn(i)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(iters):
                fn(i)
            e.record()
            torch.cuda.synchronize()
            res[mode].append(s.elapsed_time(e) / iters * 1e3)
    return min(res["0"]), min(res["1"])


# (tokens, out features, in features, batched count, bias)
SHAPES = [
    (4096, 10240, 1280, 1, True), (4096, 1280, 5120, 1, True), (4096, 3840, 1280, 1, False), (4096, 1280, 1280, 3, False),
    (4096, 1280, 1280, 1, True), (4096, 2560, 1280, 1, False),
    (16384, 5120, 640, 1, True), (16384, 640, 2560, 1, True), (16384, 1920, 640, 1, False), (16384, 640, 640, 3, False), (16384, 640, 640, 1, True),
]
print(f"{'tokens':>6s} {'out':>6s} {'in':>6s} {'x':>2s}   {'old us':>8s} {'w160 us':>8s}   {'old TF':>7s} {'w160 TF':>7s}  ratio  maxdiff/max")
for M, N, K, cnt, bias in SHAPES:
    sets = [([rb(M, N) for _ in range(cnt)], [rb(M, K) for _ in range(cnt)]) for _ in range(NSETS)]
    dws = [torch.zeros(N, K, device="cuda") for _ in range(cnt)]
    dbs = [torch.zeros(N, device="cuda") for _ in range(cnt)]
    fl = 2.0 * M * N * K * cnt

    def fn(i):
        dys, xs = sets[i % NSETS]
        if cnt == 1:
            ops.gemm_tn_f32(dys[0], xs[0], dws[0], 2, dbias=dbs[0] if bias else None)       # 2: destination known zero (the step's flat gradient buffer)
        else:
            arr = C.c_void_p * cnt
            ops.call("nk_linear_wgrad_batched", arr(*[t.data_ptr() for t in dys]), arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dws]),
                     arr(*[None] * cnt), cnt, M, N, K, N, K, K, 2, ops._stream())

    outs = {}
    for mode in ("0", "1"):
        os.environ["NK_GEMM_W160"] = mode
        for t in dws:
            t.zero_()
        fn(0)
        torch.cuda.synchronize()
        outs[mode] = dws[0].clone()
    t0, t1 = time_modes(fn)
    diff = (outs["0"] - outs["1"]).abs().max().item() / outs["0"].abs().max().item()
    print(f"{M:6d} {N:6d} {K:6d} {cnt:2d}   {t0:8.1f} {t1:8.1f}   {fl / t0 / 1e6:7.0f} {fl / t1 / 1e6:7.0f}  {t0 / t1:5.2f}  {diff:.2e}", flush=True)
