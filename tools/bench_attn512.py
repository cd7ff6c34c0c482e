"""VAE mid-block attention (single head, d = 512): the one-kernel flash forward against the two-GEMM + row-softmax path."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def rb(*shape): return torch.randn(*shape, device="cuda").to(torch.bfloat16)
for (B, L) in [(4, 16384), (1, 16384), (4, 15808), (4, 4096), (4, 1024)]:
    q, k, v = rb(B * L, 512), rb(B * L, 512), rb(B * L, 512)
    fl = 4.0 * B * L * L * 512
    t = timeit(lambda: ops.attention_fwd(q, k, v, B, 1, 512, need_lse=False))
    tu = timeit(lambda: ops.attention_unfused(q, k, v, B))
    o, o2 = ops.attention_fwd(q, k, v, B, 1, 512, need_lse=False)[0], ops.attention_unfused(q, k, v, B)
    print(f"B={B} L={L}: flash {t*1e3:8.1f} us {fl/t/1e9:6.0f} TF/s | two GEMMs + softmax {tu*1e3:8.1f} us {fl/tu/1e9:6.0f} TF/s | max diff {(o.float()-o2.float()).abs().max().item():.4f}  workgroups={B*((L+127)//128)}")
