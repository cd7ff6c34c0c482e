"""One attention shape, forward and backward, N launches each -- the subject of a rocprofv3 --pmc pass (tools/pmc_summary.py on its database).
usage: attn_one.py B H Lq Lk [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
B, H, Lq, Lk = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
D = 64
rb = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v, do = rb(B * Lq, H * D), rb(B * Lk, H * D), rb(B * Lk, H * D), rb(B * Lq, H * D)
for _ in range(iters):
    o, bwd = ops.attention_fwd(q, k, v, B, H, D)
    bwd(do)
torch.cuda.synchronize()
