"""A few representative MFMA-engine launches for rocprofv3 --pmc runs."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
for M, N, K in [(16384, 5120, 640), (4096, 10240, 1280), (16384, 640, 2560), (4096, 1280, 1280)]:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.zeros(N, K, device="cuda")
    for _ in range(3):
        ops.gemm_nt(x, w); ops.gemm_nn(dy, w); ops.gemm_tn_f32(dy, x, dw, False)
# the 3x3 convolutions of the 128^2 / 64^2 / 32^2 stages
for N, H, C in [(4, 128, 320), (4, 64, 640), (4, 32, 1280)]:
    x = ops.Img(rb(N * H * H, C), N, H, H)
    wt = torch.nn.Parameter((torch.randn(C, 3, 3, C, device="cuda") * (9 * C) ** -0.5).permute(0, 3, 1, 2))
    for _ in range(3):
        y, bwd = ops.conv2d_fwd(x, wt, None, stride=1, padding=1)
        bwd(rb(*y.t.shape))
torch.cuda.synchronize()
