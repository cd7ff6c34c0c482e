"""FeedForward.net[2] input gradient with the GEGLU backward in its epilogue (nk_linear_dgrad_geglu) against its parts: the plain input-gradient GEMM
of the same shape on the two-group kernel (default dispatch) and on the 128 x 128 double-buffer kernel the fused form runs on (NK_GEMM_G2=0), and the
stand-alone GEGLU backward kernel.   usage (GPU box): python tools/bench_geglu_dgrad.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.lib import call
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e-3
for M, N, I in [(4096, 1280, 5120), (16384, 640, 2560)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, I, device="cuda") * I ** -0.5).to(torch.bfloat16)
    u = torch.randn(M, 2 * I, device="cuda").to(torch.bfloat16)
    du = torch.empty_like(u)
    d = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * M * N * I
    t = bench(lambda: call("nk_linear_dgrad_geglu", dy.data_ptr(), w.data_ptr(), u.data_ptr(), du.data_ptr(), M, N, I, N, I, 2 * I, 2 * I, ops._stream()))
    print(f"dgrad_geglu (fused)          M={M} N={N} I={I}: {t*1e6:7.1f} us = {fl/t/1e12:5.0f} TFLOP/s")
    ts = bench(lambda: call("nk_linear_dgrad_geglu_s", dy.data_ptr(), w.data_ptr(), u.data_ptr(), du.data_ptr(), M, N, I, N, I, 2 * I, 2 * I, ops._stream()))
    print(f"dgrad_geglu_s (saved derivative: two products)         : {ts*1e6:7.1f} us = {fl/ts/1e12:5.0f} TFLOP/s")
    x = torch.randn(M, N, device="cuda").to(torch.bfloat16); w1 = (torch.randn(2 * I, N, device="cuda") * N ** -0.5).to(torch.bfloat16)
    hbuf = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
    if ops.query("nk_linear_fwd_geglu_ok", M, I, N):
        tf0 = bench(lambda: call("nk_linear_fwd_geglu", x.data_ptr(), w1.data_ptr(), None, u.data_ptr(), hbuf.data_ptr(), M, I, N, N, N, 2 * I, I, ops._stream()))
        tf1 = bench(lambda: call("nk_linear_fwd_geglu_s", x.data_ptr(), w1.data_ptr(), None, u.data_ptr(), hbuf.data_ptr(), M, I, N, N, N, 2 * I, I, ops._stream()))
        print(f"fwd_geglu (keeps u) / fwd_geglu_s (keeps s)             : {tf0*1e6:7.1f} / {tf1*1e6:7.1f} us = {2*fl/tf0/1e12:5.0f} / {2*fl/tf1/1e12:5.0f} TFLOP/s")
    t1 = bench(lambda: ops.gemm_nn(dy, w, out=d))
    print(f"plain dgrad, default kernel                     : {t1*1e6:7.1f} us = {fl/t1/1e12:5.0f} TFLOP/s")
    os.environ["NK_GEMM_G2"] = "0"
    t2 = bench(lambda: ops.gemm_nn(dy, w, out=d))
    os.environ.pop("NK_GEMM_G2")
    print(f"plain dgrad, 128x128 double-buffer kernel       : {t2*1e6:7.1f} us = {fl/t2/1e12:5.0f} TFLOP/s")
    t3 = bench(lambda: call("nk_geglu_bwd", d.data_ptr(), u.data_ptr(), du.data_ptr(), M, I, ops._stream()))
    print(f"stand-alone GEGLU backward                      : {t3*1e6:7.1f} us  ({(2 * M * I + 8 * M * I) / t3 / 1e12:.2f} TB/s over d, u, du)")
