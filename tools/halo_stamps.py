"""Where a workgroup of the halo-tile 3 x 3 convolution spends its time (diagnostic build: gemm.hip compiled with -DNK_HALO_STAMPS into
neurosis_amd/csrc/diag/libneurosis_hip_halostamps.so; NEUROSIS_HIP_LIB points the loader at it).  Per workgroup (first 8192): entry -> prologue
barrier -> k loop done -> epilogue drained, in shader cycles; the clock held in the loop; and how the workgroups of one CU follow each other."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops, lib

L = lib.load()
fn = L.nk_debug_halo_stamps
fn.argtypes = [C.c_void_p, C.c_int]
fn.restype = C.c_int
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
for (N, H, W, Ci, Co, stats) in [(4, 1024, 1024, 128, 128, 32), (4, 512, 512, 256, 256, 32), (4, 256, 256, 512, 512, 32), (4, 32, 32, 1280, 1280, None), (4, 128, 128, 320, 320, None)]:
    x = ops.Img(rb(N * H * W, Ci), N, H, W)
    wt = torch.nn.Parameter((torch.randn(Co, 3, 3, Ci, device="cuda") * 0.02).permute(0, 3, 1, 2))
    f = lambda: ops.conv2d_fwd(x, wt, None, stride=1, padding=1, stats_groups=stats)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 5 * 1e3
    nwg = 8192
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, nwg) == 0
    b = buf.reshape(nwg, 8).astype(np.int64)
    b = b[b[:, 3] > b[:, 0]]                      # workgroups that ran in the last launch (grids smaller than 8192 leave stale rows at 0)
    pro, loop, epi = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1], b[:, 3] - b[:, 2]
    rt = (b[:, 5] - b[:, 4]).astype(np.float64)
    ghz = np.median(loop / np.maximum(rt, 1) * 0.1)
    nk = (Ci // 64) * 9
    fl = 2.0 * N * H * W * Co * 9 * Ci
    tot = np.median(b[:, 3] - b[:, 0])
    print(f"{N} x {H}x{W} {Ci} -> {Co}{' +stats' if stats else ''}: {us:.1f} us/launch = {fl / us / 1e6:.0f} TFLOP/s | per workgroup, cycles median: prologue {np.median(pro):.0f}  "
          f"loop {np.median(loop):.0f} ({np.median(loop) / nk:.0f}/k-step, {nk} k-steps)  epilogue {np.median(epi):.0f}  entry->exit {tot:.0f} | clock in loop {ghz:.2f} GHz | "
          f"workgroups stamped {len(b)}; launch = {us * ghz * 1e3:.0f} cycles => {us * ghz * 1e3 / max(tot, 1):.1f} workgroup lifetimes back to back per slot", flush=True)

# per-k-step stamps of 64 workgroups in the middle of the LAST launch above is not what we want: re-run the first shape and read them
N, H, W, Ci, Co, stats = 4, 1024, 1024, 128, 128, 32
x = ops.Img(rb(N * H * W, Ci), N, H, W)
wt = torch.nn.Parameter((torch.randn(Co, 3, 3, Ci, device="cuda") * 0.02).permute(0, 3, 1, 2))
for _ in range(3): ops.conv2d_fwd(x, wt, None, stride=1, padding=1, stats_groups=stats)
torch.cuda.synchronize()
fk = L.nk_debug_halo_ksteps
fk.argtypes = [C.c_void_p]; fk.restype = C.c_int
kb = np.zeros(3 * 16 * 20 * 5, dtype=np.uint64)
assert fk(kb.ctypes.data) == 0
kb = kb.reshape(3, 16, 20, 5).astype(np.int64)
nk = (Ci // 64) * 9
print("4 x 1024x1024 128 -> 128: phases of a k-step, cycles, median over 16 workgroups and k-steps 2..15")
for wv, name in ((0, "wave 0 (group 0, stages weights)"), (1, "wave 4 (group 1, stages weights)"), (2, "wave 6 (group 1, stages the halo)")):
    a = kb[wv, :, 2:16, :]
    nxt = kb[wv, :, 3:17, :]
    dma = np.median(a[..., 1] - a[..., 0]); mf = np.median(a[..., 2] - a[..., 1]); bm = np.median(a[..., 3] - a[..., 2])
    rd = np.median(nxt[..., 4] - a[..., 3]); br = np.median(nxt[..., 0] - nxt[..., 4]); tot = np.median(nxt[..., 0] - a[..., 0])
    print(f"  {name:36s}: weight DMA issue {dma:5.0f} | MFMAs issued {mf:5.0f} | wait at M barrier {bm:5.0f} | reads + pieces + vmcnt wait {rd:5.0f} | wait at R barrier {br:5.0f} | k-step {tot:5.0f}")
