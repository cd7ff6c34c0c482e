import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
def rb(*shape): return torch.randn(*shape, device="cuda").to(torch.bfloat16)
for (B, Hh, L) in [(4, 20, 1024), (4, 10, 4096)]:
    D = 64
    q, k, v = rb(B * L, Hh * D), rb(B * L, Hh * D), rb(B * L, Hh * D)
    for _ in range(3):
        o, bwd = ops.attention_fwd(q, k, v, B, Hh, D)
        bwd(rb(B * L, Hh * D))
torch.cuda.synchronize()
