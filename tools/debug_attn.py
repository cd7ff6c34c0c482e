"""Debug aid: the adversarial head-dim-64 attention input of tests/test_kernels_gpu.py, error by row group (forward, LSE, gradients)."""
import math, os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neurosis_amd import ops
from neurosis_amd.lib import NkAttnDesc
from tests.util import bf16_round

B, H, L, D = 1, 2, 512, 64
g = torch.Generator().manual_seed(77)
q = torch.randn(B * L, H, D, generator=g) * 0.5
k = torch.randn(B * L, H, D, generator=g) * 0.5
v = torch.randn(B * L, H * D, generator=g)
q[:, :, 3] = 0.0; q[:, :, 17] = 0.0
k[:, :, 3] = 0.0; k[:, :, 17] = 0.0
q[0::4, :, 3] = 32.0
q[2::8, :, 17] = 32.0
k[:64, :, 3] = -1.5
k[130:134, :, 3] = 0.7
k[260:262, :, 3] = 1.55
if os.environ.get("PART", "2") == "1":
    k[384:416, :, 17] = 6.2 / 4
else:
    k[400:403, :, 17] = float(os.environ.get("WJ", "28")) / 4
k[470, :, 3] = 3.0
q, k = bf16_round(q.reshape(B * L, H * D)), bf16_round(k.reshape(B * L, H * D))
v = bf16_round(v)
gg = torch.Generator().manual_seed(5)
do = bf16_round(torch.randn(B * L, H * D, generator=gg))

qh, kh, vh, doh = (t.view(L, H, D).transpose(0, 1).double() for t in (q, k, v, do))
S = (qh @ kh.transpose(1, 2)) * D ** -0.5
lse_ref = torch.logsumexp(S, -1)
P = torch.softmax(S, -1)
O = P @ vh
dP = doh @ vh.transpose(1, 2)
dS = P * (dP - (doh * O).sum(-1, keepdim=True))
dq_ref = (D ** -0.5 * dS @ kh).transpose(0, 1).reshape(L, H * D)
dk_ref = (D ** -0.5 * dS.transpose(1, 2) @ qh).transpose(0, 1).reshape(L, H * D)
dv_ref = (P.transpose(1, 2) @ doh).transpose(0, 1).reshape(L, H * D)
o_ref = O.transpose(0, 1).reshape(L, H * D)

dev = lambda t: t.cuda().to(torch.bfloat16)
qd, kd, vd = dev(q), dev(k), dev(v)
o = torch.empty(L, H * D, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, H, L, dtype=torch.float32, device="cuda")
o2, bwd = ops.attention_fwd(qd, kd, vd, B, H, D)
# LSE through the C-ABI directly
d = NkAttnDesc()
d.B, d.H, d.Lq, d.Lk, d.D = B, H, L, L, D
d.sq = d.sk = d.sv = d.so = H * D
d.bq = d.bk = d.bv = d.bo = L * H * D
d.scale = D ** -0.5
ops.call("nk_attention_fwd", C.byref(d), qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), lse.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
dq, dk, dv = bwd(dev(do))
def rows(name, got, ref):
    e = (got.double().cpu() - ref).abs()
    den = ref.abs().max()
    per_row = e.max(-1).values / den
    worst = per_row.argsort(descending=True)[:6]
    print(f"{name}: max {float(per_row.max()):.3e}; worst rows {[(int(i), round(float(per_row[i]), 4)) for i in worst]}")
rows("o", o, o_ref); rows("dq", dq, dq_ref); rows("dk", dk, dk_ref); rows("dv", dv, dv_ref)
le = (lse[0].double().cpu() - lse_ref).abs()
print("lse max err", float(le.max()), "rows", [(int(i) // L, int(i) % L, round(float(le.view(-1)[i]), 4)) for i in le.view(-1).argsort(descending=True)[:6]])
