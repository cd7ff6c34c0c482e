"""us per launch of the frozen VAE encoder's convolutions (forward only) at batch 4, 1024^2."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.ops import Img
def rb(*shape, s=1.0): return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (N, HW, Ci, Co, stride) in [(4, 1024, 128, 128, 1), (4, 512, 128, 256, 1), (4, 512, 256, 256, 1), (4, 256, 256, 512, 1), (4, 256, 512, 512, 1), (4, 128, 512, 512, 1),
                                (4, 1024, 128, 128, 2), (4, 512, 256, 256, 2)]:
    x = Img(rb(N * HW * HW, Ci), N, HW, HW)
    w = torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda(), requires_grad=False)
    us = t(lambda: ops.conv2d_fwd(x, w, None, stride=stride, padding=1 if stride == 1 else 0, asym_pad=stride == 2, need_dx=False))
    Ho = HW // stride
    print(f"{N} x {HW}^2 {Ci:4d} -> {Co:4d} stride {stride}: {us:9.1f} us  {2.0 * N * Ho * Ho * Ci * Co * 9 / us / 1e6:7.0f} TFLOP/s", flush=True)
