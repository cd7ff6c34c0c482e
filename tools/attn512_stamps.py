"""Where an iteration of the head-dim-512 attention kernel spends its cycles (diagnostic build: make -C neurosis_amd/csrc EXTRA=-DNK_ATTN_STAMPS).
Per workgroup (wave 0): prologue, key loop, and inside the loop the three stretches between the points where no LDS read is outstanding:
  S  = score chain of tile t + 1 (batches 0-6) with the softmax of tile t in its shadow, up to the last K fragments' arrival
  PV = last chain batch + the second product's batches 0-6, up to the last V fragments' arrival
  sync = last four MFMAs issued, tile DMA wait, barrier"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops, lib
L_ = lib.load()
fn = L_.nk_debug_attn_stamps
fn.argtypes = [C.c_void_p, C.c_int]; fn.restype = C.c_int
def rb(*shape): return torch.randn(*shape, device="cuda").to(torch.bfloat16)
for (B, L) in [(4, 16384), (2, 16384), (1, 16384)]:
    q, k, v = rb(B * L, 512), rb(B * L, 512), rb(B * L, 512)
    f = lambda: ops.attention_fwd(q, k, v, B, 1, 512, need_lse=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 5 * 1e3
    nwg = min(8192, (L // 128) * B)
    buf = np.zeros(nwg * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, nwg) == 0
    b = buf.reshape(nwg, 8).astype(np.int64)
    pro, loop = b[:, 1] - b[:, 0], b[:, 2] - b[:, 1]
    rt = (b[:, 5] - b[:, 4]).astype(np.float64)
    ghz = np.median(loop / np.maximum(rt, 1) * 0.1)
    nt = L // 32
    print(f"B={B} L={L}: {us:.1f} us/launch, {nwg} workgroups | prologue {np.median(pro):.0f} cycles, loop {np.median(loop):.0f} = {np.median(loop) / nt:.0f}/tile "
          f"(S {np.median(b[:, 3]) / nt:.0f}  PV {np.median(b[:, 6]) / nt:.0f}  sync {np.median(b[:, 7]) / nt:.0f}; MFMA floor 2 x 1024) | clock {ghz:.2f} GHz | "
          f"loop p10/p90 {np.percentile(loop, 10) / nt:.0f}/{np.percentile(loop, 90) / nt:.0f} per tile")
