"""The halo-tile conv weight gradient (csrc/conv_wgrad_halo.h) against the gather kernel it replaces, on the step's 3 x 3 shapes."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from neurosis_amd import ops
from neurosis_amd.lib import call

SHAPES = [  # N, H, W, Cin, Cout  (UNet at batch 4 / 1024^2; VAE decoder / encoder training shapes at 256^2)
    (4, 128, 128, 320, 320), (4, 64, 64, 320, 640), (4, 64, 64, 640, 640), (4, 32, 32, 640, 1280), (4, 32, 32, 1280, 1280),
    (4, 32, 32, 2560, 1280), (4, 64, 64, 1920, 640), (4, 64, 64, 1280, 640), (4, 64, 64, 960, 640), (4, 128, 128, 960, 320), (4, 128, 128, 640, 320),
    (4, 256, 256, 128, 128), (4, 128, 128, 256, 256), (4, 64, 64, 512, 512), (4, 32, 32, 512, 512),
]


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for N, H, W, Cin, Cout in SHAPES:
    x = (torch.randn(N * H * W, Cin, device="cuda")).to(torch.bfloat16)
    dy = (torch.randn(N * H * W, Cout, device="cuda")).to(torch.bfloat16)
    dw = torch.zeros(Cout, 9 * Cin, device="cuda")
    db = torch.zeros(Cout, device="cuda")
    d = ops._conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, H, W, False)
    run = lambda: call("nk_conv2d_wgrad_bias", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, ops._stream())
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    os.environ["NK_CONV_WGRAD_HALO"] = "2"
    t_h = bench(run)
    os.environ["NK_CONV_WGRAD_HALO"] = "0"
    t_g = bench(run)
    os.environ.pop("NK_CONV_WGRAD_HALO")
    print(f"{N}x{H}x{W} {Cin:4d}->{Cout:4d}: halo {t_h * 1e6:7.1f} us = {fl / t_h / 1e12:6.0f} TFLOP/s | gather {t_g * 1e6:7.1f} us = {fl / t_g / 1e12:6.0f} | x{t_g / t_h:.2f}", flush=True)
