import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16384, 5120, 2048))]
x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16); w = (torch.randn(N, K, device="cuda") * 0.5).to(torch.bfloat16)
for _ in range(5): ops.gemm_nt(x, w)
torch.cuda.synchronize()
