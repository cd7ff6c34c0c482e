"""LayerNorm kernels alone, launched back to back through the C-ABI (no allocator, no Python op layer between launches): us per launch and
algorithmic GB/s at the two SDXL shapes.  usage (GPU box): python tools/bench_ln.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import lib
from neurosis_amd.lib import call, query

def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

st = torch.cuda.current_stream().cuda_stream
for (M, C) in [(4096, 1280), (16384, 640)]:
    # a ring of buffers larger than the Infinity Cache would be "cold"; in the step these tensors were written by the previous kernel, i.e. warm
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); dy = torch.randn_like(x); add = torch.randn_like(x); dx = torch.empty_like(x); y = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    ws = torch.empty(query("nk_layernorm_ws_floats", M, C), device="cuda")
    rows = query("nk_layernorm_part_rows", M)
    n = M * C
    t = timeit(lambda: call("nk_layernorm_fwd", x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, C, 1e-5, st))
    print(f"({M},{C}) fwd                  {t:7.1f} us  {4*n/t/1e3:7.0f} GB/s")
    t = timeit(lambda: call("nk_layernorm_bwd_dx", dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), add.data_ptr(), dx.data_ptr(), M, C, st))
    print(f"({M},{C}) bwd dx               {t:7.1f} us  {8*n/t/1e3:7.0f} GB/s")
    t = timeit(lambda: call("nk_layernorm_bwd_params", dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), M, C, 0, st))
    print(f"({M},{C}) bwd params + reduce  {t:7.1f} us")
    t = timeit(lambda: call("nk_layernorm_bwd_rows", dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), add.data_ptr(), dx.data_ptr(), ws.data_ptr(), M, C, st))
    print(f"({M},{C}) bwd rows (one pass)  {t:7.1f} us  {8*n/t/1e3:7.0f} GB/s   ({rows} partial rows)")
    bt = lib.NkColpartBatch()
    for z in range(3):
        bt.part[z], bt.dgamma[z], bt.dbeta[z], bt.nrows[z], bt.C[z], bt.accumulate[z] = ws.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, 0
    bt.n = 3
    import ctypes
    t = timeit(lambda: call("nk_colpart_reduce_batch", ctypes.byref(bt), st))
    print(f"({M},{C}) reduce of 3 LNs      {t:7.1f} us")
