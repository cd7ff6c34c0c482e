"""The dispatcher-visible ops (torch.ops.neurosis_hip.*) run the same HIP kernels as the explicit chain and differentiate
through torch autograd: values and gradients against fp32 PyTorch on the same (bf16-rounded) inputs.  Tolerances: bf16 outputs
2e-2 of the max magnitude, fp32 weight gradients 1e-2."""
import pytest
import torch
import torch.nn.functional as F

import neurosis_amd.torch_ops  # noqa: F401  (registers the namespace)
from tests.util import rel_err

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
o = torch.ops.neurosis_hip


def rnd(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).to(bf)


def test_linear_layernorm_geglu_chain_through_autograd():
    x, w, b = rnd(512, 256, seed=1), rnd(1024, 256, seed=2, scale=0.06), torch.randn(1024, generator=torch.Generator().manual_seed(3)) * 0.1
    g, be = 1 + 0.1 * torch.randn(256, generator=torch.Generator().manual_seed(4)), 0.1 * torch.randn(256, generator=torch.Generator().manual_seed(5))
    xc, wc = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bc, gc, bec = b.cuda().requires_grad_(True), g.cuda().requires_grad_(True), be.cuda().requires_grad_(True)
    y = o.geglu(o.linear(o.layernorm(xc, gc, bec, 1e-5), wc, bc))
    up = rnd(512, 512, seed=6)
    y.backward(up.cuda())
    xr, wr, br, gr, ber = (t.float().requires_grad_(True) for t in (x, w, b, g, be))
    u = F.linear(F.layer_norm(xr, (256,), gr, ber, 1e-5).to(bf).float(), wr, br).to(bf).float()
    yr = u[:, :512] * F.gelu(u[:, 512:])
    yr.backward(up.float())
    assert rel_err(y.float(), yr) <= 2e-2
    assert rel_err(xc.grad.float(), xr.grad) <= 3e-2 and rel_err(wc.grad.float(), wr.grad) <= 2e-2
    assert rel_err(bc.grad, br.grad) <= 2e-2 and rel_err(gc.grad, gr.grad) <= 2e-2 and rel_err(bec.grad, ber.grad) <= 2e-2


def test_attention_conv_groupnorm_ops_through_autograd():
    B, H, L, Lk, D = 2, 4, 256, 77, 64
    q, k, v = rnd(B * L, H * D, seed=1), rnd(B * Lk, H * D, seed=2), rnd(B * Lk, H * D, seed=3)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = o.attention(qc, kc, vc, B, H)
    up = rnd(B * L, H * D, seed=4)
    out.backward(up.cuda())
    sp = lambda t, n: t.float().view(B, n, H, D).transpose(1, 2)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    ref = F.scaled_dot_product_attention(sp(qr, L), sp(kr, Lk), sp(vr, Lk)).transpose(1, 2).reshape(B * L, H * D)
    ref.backward(up.float())
    assert rel_err(out.float(), ref) <= 2e-2
    for a, r in ((qc, qr), (kc, kr), (vc, vr)):
        assert rel_err(a.grad.float(), r.grad) <= 3e-2
    # conv (3x3, stride 2) + GroupNorm+SiLU on channels-last tokens
    N, Hh, Ww, Ci, Co = 2, 16, 16, 32, 64
    x = rnd(N, Ci, Hh, Ww, seed=5)
    w = rnd(Co, Ci, 3, 3, seed=6, scale=0.06)
    gam, bet = 1 + 0.1 * torch.randn(Co, generator=torch.Generator().manual_seed(7)), 0.1 * torch.randn(Co, generator=torch.Generator().manual_seed(8))
    xt = x.permute(0, 2, 3, 1).reshape(N * Hh * Ww, Ci).contiguous().cuda().requires_grad_(True)
    wt = w.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)            # [Cout, KH, KW, Cin]
    gc, bc = gam.cuda().requires_grad_(True), bet.cuda().requires_grad_(True)
    y = o.groupnorm_silu(o.conv2d(xt, wt, None, N, Hh, Ww, 2, 1), gc, bc, N, 32, 1e-5, True)
    upc = rnd(N * 8 * 8, Co, seed=9)
    y.backward(upc.cuda())
    xr, wr, gr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    c = F.conv2d(xr, wr, None, stride=2, padding=1).to(bf).float()
    yr = F.silu(F.group_norm(c, 32, gr, br, 1e-5))
    yr.backward(upc.float().view(N, 8, 8, Co).permute(0, 3, 1, 2))
    assert rel_err(y.float(), yr.permute(0, 2, 3, 1).reshape(-1, Co)) <= 2e-2
    assert rel_err(xt.grad.float(), xr.grad.permute(0, 2, 3, 1).reshape(-1, Ci)) <= 3e-2
    assert rel_err(wt.grad.float(), wr.grad.permute(0, 2, 3, 1)) <= 2e-2
    assert rel_err(gc.grad, gr.grad) <= 2e-2 and rel_err(bc.grad, br.grad) <= 2e-2
