"""The halo-tile 3x3 convolution (csrc/conv_halo.h) against the implicit-GEMM gather kernels it replaces (NK_CONV_HALO=0), forward only:
us per launch, TFLOP/s, and the largest difference between the two outputs (different accumulation order: not bit-equal).
Shapes: the frozen VAE encoder's stride-1 convolutions and the UNet's, batch 4 at 1024^2."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.ops import Img


def rb(*shape, s=1.0):
    return (torch.randn(*shape, device="cuda") * s).to(torch.bfloat16)


def t(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


SHAPES = [  # N, H, W, Cin, Cout
    (4, 1024, 1024, 128, 128), (4, 512, 512, 128, 256), (4, 512, 512, 256, 256), (4, 256, 256, 256, 512), (4, 256, 256, 512, 512), (4, 128, 128, 512, 512),
    (4, 32, 32, 1280, 1280), (4, 32, 32, 2560, 1280), (4, 64, 64, 640, 640), (4, 64, 64, 1280, 640), (4, 64, 64, 320, 640), (4, 128, 128, 320, 320), (4, 128, 128, 640, 320),
    (4, 152, 104, 320, 320), (4, 76, 52, 640, 640),
]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    SHAPES = [SHAPES[0], SHAPES[5], SHAPES[6], SHAPES[11]]
for (N, H, W, Ci, Co) in SHAPES:
    x = Img(rb(N * H * W, Ci), N, H, W)
    w = torch.nn.Parameter(ops.conv_weight_param(Co, Ci, 3, 3).data.normal_(0, (9 * Ci) ** -0.5).cuda(), requires_grad=False)
    bias = torch.randn(Co, device="cuda")
    res = rb(N * H * W, Co)
    out, us = {}, {}
    for mode, env in (("gather", {"NK_CONV_HALO": "0"}), ("halo", {})):
        os.environ.update(env)
        out[mode] = ops.conv2d_fwd(x, w, bias, residual=res, need_dx=False)[0].t.float()
        us[mode] = t(lambda: ops.conv2d_fwd(x, w, bias, residual=res, need_dx=False))
        # the output's GroupNorm sums: a statistics pass behind the convolution vs its statistics epilogue
        us[mode + "+gn"] = t(lambda: ops.groupnorm_sums(ops.conv2d_fwd(x, w, bias, residual=res, need_dx=False, stats_groups=32)[0], 32))
        for k in env:
            os.environ.pop(k, None)
    fl = 2.0 * N * H * W * Ci * Co * 9
    d = float((out["gather"] - out["halo"]).abs().max())
    ref = float(out["gather"].abs().max())
    tf = lambda u: fl / u / 1e6
    print(f"{N} x {H}x{W} {Ci:4d} -> {Co:4d}: gather {us['gather']:8.1f} us {tf(us['gather']):5.0f} TF/s | halo {us['halo']:8.1f} us"
          f" {tf(us['halo']):5.0f} TF/s | conv + output sums: statistics pass {us['gather+gn']:8.1f} us, epilogue {us['halo+gn']:8.1f} us | max diff {d:.3g} of {ref:.3g}", flush=True)
