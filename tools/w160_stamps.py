"""Where a launch of the 160-row weight-gradient kernel spends its time (diagnostic build: the library under gpurun_out/w160_stamps/, built by
`make -C neurosis_amd/csrc clean all EXTRA=-DNK_W160_STAMPS`; NK_LIB points the loader at it).  Per workgroup: compute wave 0's entry -> barrier 0 ->
k loop done -> epilogue drained in shader cycles, the clock held in the loop, and producer wave 4's split of the loop into DMA issue, vmcnt
wait and barrier wait."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops, lib

L = lib.load()
fn = L.nk_debug_w160_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
for (M, N, K) in [(4096, 1280, 5120), (4096, 10240, 1280), (4096, 3840, 1280)]:
    x, dy = rb(M, K), rb(M, N)
    dw = torch.empty(N, K, device="cuda")
    f = lambda: ops.gemm_tn_f32(dy, x, dw, False)
    for _ in range(20): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    bn = 160 if K % 160 == 0 and (N // 160) * (K // 160) >= 200 else 128
    nwg = min(4096, (N // 160) * ((K + bn - 1) // bn))
    buf = np.zeros(nwg * 12, dtype=np.uint64)
    assert fn(buf.ctypes.data, nwg) == 0
    b = buf.reshape(nwg, 12).astype(np.int64)
    pro, loop, epi = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2]
    rt = (b[:, 5] - b[:, 4]).astype(np.float64)          # 100 MHz ticks
    ghz = np.median(loop / np.maximum(rt, 1) * 0.1)
    t0 = b[:, 0].min()
    nk = M // 64
    print(f"wgrad tokens {M} out {N} in {K}: {us:.1f} us/launch, {nwg} workgroups | compute wave 0, cycles median: prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} "
          f"({np.median(loop) / nk:.0f}/slab, {nk} slabs; p10 {np.percentile(loop, 10) / nk:.0f} p90 {np.percentile(loop, 90) / nk:.0f})  epilogue {np.median(epi):.0f} | clock in loop {ghz:.2f} GHz | "
          f"first entry -> last exit {(b[:, 3].max() - t0)} cycles; entry spread {(b[:, 0].max() - t0)} | producer wave 4 per slab: issue {np.median(b[:, 6]) / nk:.0f}  "
          f"vmcnt wait {np.median(b[:, 7]) / nk:.0f}  barrier wait {np.median(b[:, 8]) / nk:.0f}", flush=True)
