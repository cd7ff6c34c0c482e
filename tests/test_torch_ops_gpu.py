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


def test_upsample_cat_ops_through_autograd():
    """openaimodel.py:126-143 (Upsample: nearest x2 + 3 x 3 conv) and :836 (torch.cat on channels): values, and the adjoints
    (2 x 2 sum-pool; channel split) against torch autograd"""
    N, H, W, Ca, Cb, Co = 2, 8, 12, 64, 32, 128
    a, b = rnd(N * H * W, Ca, seed=1), rnd(N * H * W, Cb, seed=2)
    ac, bc = a.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    cat = o.cat_channels(ac, bc)
    assert torch.equal(cat.cpu(), torch.cat([a, b], 1))
    up = rnd(N * H * W, Ca + Cb, seed=3)
    cat.backward(up.cuda())
    assert torch.equal(ac.grad.cpu(), up[:, :Ca]) and torch.equal(bc.grad.cpu(), up[:, Ca:])
    w = rnd(Co, 3, 3, Ca, seed=4, scale=0.05)
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(5)) * 0.1
    y = o.upsample2x_nearest_conv(a.cuda(), w.cuda(), bias.cuda(), N, H, W)
    xr = a.float().view(N, H, W, Ca).permute(0, 3, 1, 2).requires_grad_(True)
    upx = F.interpolate(xr, scale_factor=2, mode="nearest")
    yr = F.conv2d(upx, w.float().permute(0, 3, 1, 2), bias, padding=1)
    assert rel_err(y.float().cpu(), yr.permute(0, 2, 3, 1).reshape(-1, Co)) <= 2e-2
    dup = rnd(N * 4 * H * W, Ca, seed=6)
    dx = o.upsample2x_nearest_bwd(dup.cuda(), N, H, W)
    upx.backward(dup.float().view(N, 2 * H, 2 * W, Ca).permute(0, 3, 1, 2))
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1).reshape(-1, Ca)) <= 1e-2


def test_edm_loss_ops_through_autograd():
    """StandardDiffusionLoss (loss.py:117-157) with the EDM denoiser scalings (denoiser.py:41-53): z_t and the network input, the weighted
    per-sample loss, and d loss / d net_out through torch autograd; against fp32 torch on the bf16-rounded network output"""
    B, Cc, H, W, cpad = 3, 4, 16, 24, 8
    g = torch.Generator().manual_seed(2)
    x, eps = torch.randn(B, Cc, H, W, generator=g), torch.randn(B, Cc, H, W, generator=g)
    sig = torch.tensor([0.3, 1.7, 9.0])
    sd = 0.5
    c_skip, c_out, c_in = sd ** 2 / (sig ** 2 + sd ** 2), sig * sd / (sig ** 2 + sd ** 2) ** 0.5, 1 / (sig ** 2 + sd ** 2) ** 0.5
    wgt = (sig ** 2 + sd ** 2) / (sig * sd) ** 2
    zt, net_in = o.edm_prepare(x.cuda(), eps.cuda(), sig.cuda(), c_in.cuda(), cpad)
    zr = x + sig.view(B, 1, 1, 1) * eps
    assert rel_err(zt.cpu(), zr) <= 1e-6
    nin = (zr * c_in.view(B, 1, 1, 1)).permute(0, 2, 3, 1).reshape(-1, Cc)
    assert rel_err(net_in[:, :Cc].float().cpu(), nin) <= 1e-2 and float(net_in[:, Cc:].abs().max()) == 0.0
    net_out = rnd(B * H * W, cpad, seed=7)
    nc = net_out.cuda().requires_grad_(True)
    loss = o.edm_loss(nc, zt, x.cuda(), c_out.cuda(), c_skip.cuda(), wgt.cuda())
    coef = torch.tensor([0.2, 1.0, -0.7])
    (loss * coef.cuda()).sum().backward()
    nr = net_out.float().requires_grad_(True)
    pred = nr[:, :Cc].view(B, H, W, Cc).permute(0, 3, 1, 2) * c_out.view(B, 1, 1, 1) + zr * c_skip.view(B, 1, 1, 1)
    lr = (wgt.view(B, 1, 1, 1) * (pred - x) ** 2).flatten(1).mean(1)
    (lr * coef).sum().backward()
    assert rel_err(loss.cpu(), lr) <= 1e-4
    assert rel_err(nc.grad.float().cpu()[:, :Cc], nr.grad[:, :Cc]) <= 1e-2 and float(nc.grad[:, Cc:].abs().max()) == 0.0


def test_flat_allreduce_ops_single_rank_and_fused_round3_ops():
    """flat_allreduce_{start,wait} are the identity on one rank (no process group: the reducer's world is 1; two ranks are covered by
    tests/test_dp_gloo.py and tests/test_dp_gpu.py through the same FlatGradReducer); conv2d_fwd_stats and linear_dgrad_geglu against torch"""
    flat = torch.randn(4096, device="cuda")
    keep = flat.clone()
    o.flat_allreduce_start(flat, 0, 2048)
    o.flat_allreduce_start(flat, 2048, 4096)
    o.flat_allreduce_wait(flat)
    assert torch.equal(flat, keep)
    N, H, W, Ci, Co, G = 2, 32, 32, 64, 128, 32
    x, w = rnd(N * H * W, Ci, seed=1), rnd(Co, 3, 3, Ci, seed=2, scale=0.05)
    y, sums = o.conv2d_fwd_stats(x.cuda(), w.cuda(), None, N, H, W, G)
    yr = F.conv2d(x.float().view(N, H, W, Ci).permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1)
    assert rel_err(y.float().cpu(), yr.permute(0, 2, 3, 1).reshape(-1, Co)) <= 2e-2
    yg = y.float().cpu().view(N, H * W, G, Co // G)
    want = torch.stack([yg.sum((1, 3)), (yg * yg).sum((1, 3))], -1).reshape(N, 2 * G)
    assert rel_err(sums.cpu(), want) <= 1e-4
    M, Nn, I = 512, 320, 1280
    dy, wl, u = rnd(M, Nn, seed=3), rnd(Nn, I, seed=4, scale=0.05), rnd(M, 2 * I, seed=5)
    du = o.linear_dgrad_geglu(dy.cuda(), wl.cuda(), u.cuda())
    ur = u.float().requires_grad_(True)
    gl = ur[:, :I] * F.gelu(ur[:, I:])
    gl.backward((dy.float() @ wl.float()).to(bf).float())
    assert rel_err(du.float().cpu(), ur.grad) <= 2e-2
