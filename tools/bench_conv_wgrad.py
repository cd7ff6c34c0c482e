"""us per launch of the UNet's 3x3 convolution weight gradients (batch 4)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.ops import Img
def rb(*shape, s=1.0): return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ops.state.wgrad_stream = None
for (N, HW, Ci, Co) in [(4, 32, 1280, 1280), (4, 32, 2560, 1280), (4, 64, 640, 640), (4, 64, 1280, 640), (4, 128, 320, 320), (4, 128, 640, 320), (4, 128, 960, 320)]:
    x = Img(rb(N * HW * HW, Ci), N, HW, HW)
    w = torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda())
    dy = rb(N * HW * HW, Co)
    y, bwd = ops.conv2d_fwd(x, w, None, need_dx=False)
    us = t(lambda: bwd(dy))
    print(f"{N} x {HW}^2 {Ci:4d} -> {Co:4d}: wgrad {us:8.1f} us  {2.0 * N * HW * HW * Ci * Co * 9 / us / 1e6:6.0f} TFLOP/s", flush=True)
