"""A few launches of the head-dim-512 attention forward at BASELINE size (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
B, L = 4, 16384
q, k, v = (torch.randn(B * L, 512, device="cuda").to(torch.bfloat16) for _ in range(3))
for _ in range(3): ops.attention_fwd(q, k, v, B, 1, 512, need_lse=False)
torch.cuda.synchronize()
