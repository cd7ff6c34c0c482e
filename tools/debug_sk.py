import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
torch.set_printoptions(linewidth=250, edgeitems=40)
x = torch.eye(128, device="cuda", dtype=torch.bfloat16)
n_idx = torch.arange(128, device="cuda", dtype=torch.float32)
wn = n_idx[:, None].expand(128, 128).contiguous().to(torch.bfloat16)   # w[n][k] = n
wk = n_idx[None, :].expand(128, 128).contiguous().to(torch.bfloat16)   # w[n][k] = k
yn = ops.gemm_nt(x, wn).float()   # y[m][n] = n
ym = ops.gemm_nt(x, wk).float()   # y[m][n] = m
print("col map row0 :", yn[0].int().tolist())
print("col map row17:", yn[17].int().tolist())
print("row map col0 :", ym[:, 0].int().tolist())
print("row map col37:", ym[:, 37].int().tolist())
dy = torch.eye(128, device="cuda", dtype=torch.bfloat16)
xk = n_idx[None, :].expand(128, 128).contiguous().to(torch.bfloat16)   # x[m][k] = k
xm = n_idx[:, None].expand(128, 128).contiguous().to(torch.bfloat16)   # x[m][k] = m
dw = torch.zeros(128, 128, device="cuda")
ops.gemm_tn_f32(dy, xk, dw, False)
print("f32 col map row0:", dw[0].int().tolist())
print("f32 col map row33:", dw[33].int().tolist())
ops.gemm_tn_f32(dy, xm, dw, False)
print("f32 row map col0:", dw[:, 0].int().tolist())
