"""Microbenchmark of the MFMA tile engine on the SDXL training-step shapes (run on the GPU box)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)

rows = []
LIN = [  # (M, N, K, label)
    (16384, 640, 640, "L1 to_out/proj"), (16384, 1920, 640, "L1 qkv"), (16384, 5120, 640, "L1 ff1"), (16384, 640, 2560, "L1 ff2"),
    (4096, 1280, 1280, "L2 to_out/q"), (4096, 3840, 1280, "L2 qkv"), (4096, 10240, 1280, "L2 ff1"), (4096, 1280, 5120, "L2 ff2"),
    (308, 2560, 2048, "L2 ctx kv"), (308, 1280, 2048, "L1 ctx kv"),
]
for M, N, K, lab in LIN:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.zeros(N, K, device="cuda")
    fl = 2.0 * M * N * K
    t = timeit(lambda: ops.gemm_nt(x, w)); rows.append((lab, "fwd", M, N, K, t, fl / t / 1e9))
    t = timeit(lambda: ops.gemm_nn(dy, w)); rows.append((lab, "dgrad", M, N, K, t, fl / t / 1e9))
    t = timeit(lambda: ops.gemm_tn_f32(dy, x, dw, False)); rows.append((lab, "wgrad", M, N, K, t, fl / t / 1e9))

CONV = [  # (N,H,W,Cin,Cout,stride,ups,label)
    (4, 128, 128, 320, 320, 1, False, "c 320 128^2"), (4, 64, 64, 640, 640, 1, False, "c 640 64^2"), (4, 32, 32, 1280, 1280, 1, False, "c 1280 32^2"),
    (4, 32, 32, 2560, 1280, 1, False, "c 2560->1280 32^2"), (4, 128, 128, 960, 320, 1, False, "c 960->320 128^2"), (4, 128, 128, 320, 320, 2, False, "down 320"),
    (4, 64, 64, 640, 640, 1, True, "up 640"), (4, 1024, 1024, 128, 128, 1, False, "vae 128 1024^2"), (4, 512, 512, 256, 256, 1, False, "vae 256 512^2"),
]
for N, H, W, Ci, Co, s, up, lab in CONV:
    x = ops.Img(rb(N * H * W, Ci), N, H, W)
    wt = torch.nn.Parameter((torch.randn(Co, 3, 3, Ci, device="cuda") * 0.02).permute(0, 3, 1, 2))
    out, bwd = ops.conv2d_fwd(x, wt, None, stride=s, padding=1, upsample=up)
    fl = 2.0 * out.t.shape[0] * Co * 9 * Ci
    t = timeit(lambda: ops.conv2d_fwd(x, wt, None, stride=s, padding=1, upsample=up), 5); rows.append((lab, "fwd", out.t.shape[0], Co, 9 * Ci, t, fl / t / 1e9))
    if "vae" in lab: continue
    dy = rb(*out.t.shape)
    d = ops._conv_desc(N, H, W, Ci, Co, 3, 3, s, 1, 1, out.H, out.W, up)
    import ctypes as C
    Hin, Win = (2 * H, 2 * W) if up else (H, W)
    dx = torch.empty(N * Hin * Win, Ci, device="cuda", dtype=torch.bfloat16)
    gw = torch.zeros(Co, 9 * Ci, device="cuda")
    w2 = ops.w2d(wt)
    t = timeit(lambda: ops.call("nk_conv2d_dgrad", C.byref(d), dy.data_ptr(), w2.data_ptr(), dx.data_ptr(), ops._stream()), 5); rows.append((lab, "dgrad", 0, 0, 0, t, fl / t / 1e9))
    t = timeit(lambda: ops.call("nk_conv2d_wgrad", C.byref(d), dy.data_ptr(), x.t.data_ptr(), gw.data_ptr(), 0, ops._stream()), 5); rows.append((lab, "wgrad", 0, 0, 0, t, fl / t / 1e9))

for r in rows:
    print(f"{r[0]:22s} {r[1]:6s} M={r[2]:<8d} N={r[3]:<6d} K={r[4]:<6d} {r[5]*1e3:9.1f} us {r[6]:8.1f} TF/s")
