"""us per launch of the UNet's 3x3 convolutions (forward and input gradient), with the two-group ring kernel off / by shape / forced
(NK_GEMM_G2 = 0 / 1 / 2), and the values of the forced run against the kernels it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.ops import Img
def rb(*shape, s=1.0): return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)
def t(fn, iters=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
shapes = [(4, 32, 1280, 1280), (4, 32, 2560, 1280), (4, 32, 1920, 1280), (4, 64, 640, 640), (4, 64, 1280, 640), (4, 64, 1920, 640), (4, 64, 960, 640),
          (4, 128, 320, 320), (4, 128, 640, 320), (4, 128, 960, 320), (4, 64, 320, 640), (4, 32, 640, 1280)]
print(f"{'N x HW, Cin -> Cout':26s} {'fwd 0':>8s} {'fwd 1':>8s} {'fwd 2':>8s} | {'dgrad 0':>8s} {'dgrad 1':>8s} {'dgrad 2':>8s}   max |diff| fwd / dgrad (forced vs off)")
for (N, HW, Ci, Co) in shapes:
    x = Img(rb(N * HW * HW, Ci), N, HW, HW)
    w = torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda())
    dy = rb(N * HW * HW, Co)
    res, outs = {}, {}
    for mode in ("0", "1", "2"):
        os.environ["NK_GEMM_G2"] = mode
        y, bwd = ops.conv2d_fwd(x, w, None)
        ops.state.wgrad_stream = None
        dx = bwd(dy)[0]
        outs[mode] = (y.t.float().clone(), dx.t.float().clone())
        flops = 2.0 * N * HW * HW * Ci * Co * 9
        tf = t(lambda: ops.conv2d_fwd(x, w, None, need_dx=False))
        # dgrad alone: time bwd and subtract the weight gradient's share by timing it with a frozen weight
        w.requires_grad_(False)
        y2, bwd2 = ops.conv2d_fwd(x, w, None)
        td = t(lambda: bwd2(dy))
        w.requires_grad_(True)
        res[mode] = (tf, td, flops / tf / 1e6, flops / td / 1e6)
    os.environ.pop("NK_GEMM_G2", None)
    ef = float((outs["2"][0] - outs["0"][0]).abs().max()); ed = float((outs["2"][1] - outs["0"][1]).abs().max())
    print(f"{N} x {HW}^2, {Ci:4d} -> {Co:4d}      " + " ".join(f"{res[m][0]:8.1f}" for m in "012") + " | " + " ".join(f"{res[m][1]:8.1f}" for m in "012")
          + f"   {ef:.3g} / {ed:.3g}   TF/s fwd {res['0'][2]:.0f} -> {res['2'][2]:.0f}, dgrad {res['0'][3]:.0f} -> {res['2'][3]:.0f}", flush=True)
