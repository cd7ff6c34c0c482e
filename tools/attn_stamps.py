"""Where a head-dim-64 attention forward launch spends its time (diagnostic build: make -C neurosis_amd/csrc clean all EXTRA=-DNK_ATTN_STAMPS).
Per workgroup: entry -> prologue done -> key loop done -> epilogue drained, shader cycles; clock held in the loop; spread of entries / exits."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops, lib
L_ = lib.load()
fn = L_.nk_debug_attn_stamps
fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
def rb(*shape): return torch.randn(*shape, device="cuda").to(torch.bfloat16)
for (B, H, Lq, Lk) in [(4, 20, 1024, 1024), (4, 10, 4096, 4096), (4, 20, 1024, 77), (16, 20, 1024, 1024)]:
    D = 64
    q, k, v = rb(B * Lq, H * D), rb(B * Lk, H * D), rb(B * Lk, H * D)
    f = lambda: ops.attention_fwd(q, k, v, B, H, D)
    for _ in range(10): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    nwg = min(8192, (Lq // 128) * H * B)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, nwg) == 0
    b = buf.reshape(nwg, 8).astype(np.int64)
    pro, loop, epi = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2]
    rt = (b[:, 5] - b[:, 4]).astype(np.float64)
    ghz = np.median(loop / np.maximum(rt, 1) * 0.1)
    nt = (Lk + 63) // 64
    t0 = b[:, 0].min()
    print(f"B={B} H={H} Lq={Lq} Lk={Lk}: {us:.1f} us/launch, {nwg} workgroups | cycles median: prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} ({np.median(loop) / nt:.0f}/tile)  "
          f"epilogue {np.median(epi):.0f} | clock {ghz:.2f} GHz | entries spread over {(b[:, 0].max() - t0)} cycles, last exit at {(b[:, 3].max() - t0)}; "
          f"loop p10/p90 {np.percentile(loop, 10):.0f}/{np.percentile(loop, 90):.0f}")
