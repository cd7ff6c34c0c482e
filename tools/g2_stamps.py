"""Where a two-group GEMM launch spends its time (diagnostic build: make -C neurosis_amd/csrc clean all EXTRA=-DNK_G2_STAMPS).
Per workgroup: entry -> prologue done -> k loop done -> epilogue drained, in shader cycles, and the clock held in the loop."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops, lib

L = lib.load()
fn = L.nk_debug_g2_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
os.environ["NK_GEMM_G2"] = "2"
for (M, N, K, kind) in [(4096, 1280, 1280, "fwd"), (4096, 1280, 1280, "dgrad"), (4096, 1280, 5120, "fwd"), (4096, 3840, 1280, "fwd"), (16384, 640, 640, "fwd"),
                        (4096, 3840, 1280, "dgrad"), (4096, 3840, 1280, "wgrad"), (4096, 1280, 5120, "wgrad"), (4096, 10240, 1280, "wgrad")]:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.empty(N, K, device="cuda")
    f = (lambda: ops.gemm_nt(x, w)) if kind == "fwd" else ((lambda: ops.gemm_nn(dy, w)) if kind == "dgrad" else (lambda: ops.gemm_tn_f32(dy, x, dw, False)))
    for _ in range(20): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    # tiles and k-steps: forward [M x N] over K; dgrad [M x K] over N; wgrad [N x K] over M
    rows, cols, red = (M, N, K) if kind == "fwd" else ((M, K, N) if kind == "dgrad" else (N, K, M))
    nwg = min(4096, (rows // 128) * (cols // 160))
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, nwg) == 0
    b = buf.reshape(nwg, 8).astype(np.int64)
    pro, loop, epi = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2]
    rt = (b[:, 5] - b[:, 4]).astype(np.float64)          # 100 MHz ticks
    ghz = np.median(loop / np.maximum(rt, 1) * 0.1)
    t0 = b[:, 0].min()
    print(f"{kind} {M}x{N}x{K}: {us:.1f} us/launch | cycles median: prologue {np.median(pro):.0f}  loop {np.median(loop):.0f} ({np.median(loop) / (red // 64):.0f}/k-step, {red // 64} k-steps, {nwg} tiles)  "
          f"epilogue {np.median(epi):.0f} | clock in loop {ghz:.2f} GHz | first entry -> last exit {(b[:, 3].max() - t0)} cycles; entry spread {(b[:, 0].max() - t0)}")
