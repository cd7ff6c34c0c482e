"""A few representative MFMA-engine launches for rocprofv3 --pmc runs."""
import sys, torch
sys.path.insert(0, ".")
from neurosis_amd import ops
def rb(*shape): return (torch.randn(*shape, device="cuda") * 0.5).to(torch.bfloat16)
for M, N, K in [(16384, 5120, 640), (4096, 10240, 1280), (16384, 640, 2560), (4096, 1280, 1280)]:
    x, w, dy = rb(M, K), rb(N, K), rb(M, N)
    dw = torch.zeros(N, K, device="cuda")
    for _ in range(3):
        ops.gemm_nt(x, w); ops.gemm_nn(dy, w); ops.gemm_tn_f32(dy, x, dw, False)
torch.cuda.synchronize()
