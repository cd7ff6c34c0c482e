"""VERDICT round 5, item 4, the numerics half: LayerNorm folded algebraically into the GEMM that consumes it,
    y = LN(x) W^T + b   =   rstd_i (x_i . W'_n  -  mu_i s_n) + b'_n,     W' = bf16(W gamma),  s_n = sum_k W'_nk,  b'_n = sum_k beta_k W_nk + b_n,
emulated in PyTorch (fp32 accumulation over bf16 operands, as the MFMA does) on the REAL residual stream of the benchmark's full-depth SDXL UNet:
the inputs of norm1 / norm2 / norm3 of transformer blocks at several depths are captured in one forward pass, and for each the error of
  (a) today's path: bf16(LN(x)) times bf16 W, fp32 accumulation, bf16 output, and
  (b) the folded path: bf16 x times bf16 W', fp32 accumulation, the correction in fp32, bf16 output
against the fp64 evaluation of the exact formula is reported (relative to the largest output magnitude), with the stream's |mu| / sigma; then the same
on the captured rows with their mean inflated to |mu| / sigma = 10, 30, 100 (what a trained network's residual stream can look like).
    python tools/ln_fusion_numerics.py            # GPU box
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from neurosis_amd import ops  # noqa: E402
from neurosis_amd.modules.attention import BasicTransformerBlock  # noqa: E402

dev = torch.device("cuda", 0)
eng = bench.build_engine(dev, (1024, 1024), None)
unet = eng.model.diffusion_model
blocks = [(n, m) for n, m in unet.named_modules() if isinstance(m, BasicTransformerBlock)]
want = [blocks[0], blocks[3], blocks[len(blocks) // 2], blocks[-12], blocks[-1]]
captured = []
orig = ops.layernorm_fwd


def spy(x, w, b, eps, *a, **k):
    for name, blk in want:
        for which, ln in (("norm1", blk.norm1), ("norm2", blk.norm2), ("norm3", blk.norm3)):
            if ln.weight is w:
                captured.append((f"{name}.{which}", x.detach().clone(), w.detach().float().clone(), b.detach().float().clone(), eps, blk, which))
    return orig(x, w, b, eps, *a, **k)


ops.layernorm_fwd = spy
import neurosis_amd.modules.attention as A  # noqa: E402

A.ops.layernorm_fwd = spy
gen = torch.Generator(device=dev).manual_seed(42)
batch = bench.synthetic_batch(dev, 4, (1024, 1024), gen, True)
with torch.no_grad():
    eng.training_step(batch, 0, sigmas=bench.draw_sigmas(4, gen, dev))
torch.cuda.synchronize()
ops.layernorm_fwd = orig
A.ops.layernorm_fwd = orig


def consumer_weight(blk, which):
    if which == "norm1":
        a = blk.attn1
        w = getattr(a, "to_qkv", None)
        return (w.weight if w is not None else a.to_q.weight).detach(), None
    if which == "norm2":
        return blk.attn2.to_q.weight.detach(), None
    proj = blk.ff.net[0].proj
    return proj.weight.detach(), proj.bias.detach()


def bf(t):
    return t.to(torch.bfloat16).float()


def compare(tag, x, gamma, beta, eps, W, bias):
    xf = x.float()                                   # x is bf16 already (the residual stream as stored)
    rows = torch.randperm(xf.shape[0], device=dev)[:1024]
    xf = xf[rows]
    Wf = W.float()
    N = min(Wf.shape[0], 2560)
    Wf = Wf[:N]
    bias_f = bias.float()[:N] if bias is not None else torch.zeros(N, device=dev)
    x64, W64, g64, b64 = xf.double(), Wf.double(), gamma.double(), beta.double()
    mu64 = x64.mean(-1, keepdim=True)
    var64 = x64.var(-1, unbiased=False, keepdim=True)
    true = ((x64 - mu64) / (var64 + eps).sqrt() * g64 + b64) @ W64.t() + bias_f.double()
    scale = float(true.abs().max())
    # (a) today's path
    mu, var = xf.mean(-1, keepdim=True), xf.var(-1, unbiased=False, keepdim=True)
    rstd = (var + eps).rsqrt()
    y = bf((xf - mu) * rstd * gamma + beta)
    out_a = bf(y @ Wf.t() + bias_f)
    # (b) folded
    Wp = bf(Wf * gamma)
    s = Wp.sum(-1)
    bp = Wf @ beta + bias_f
    acc = xf @ Wp.t()
    out_b = bf(rstd * (acc - mu * s) + bp)
    ea = float((out_a.double() - true).abs().max()) / scale
    eb = float((out_b.double() - true).abs().max()) / scale
    ra = float((out_a.double() - true).pow(2).mean().sqrt()) / scale
    rb = float((out_b.double() - true).pow(2).mean().sqrt()) / scale
    ratio = float((mu.abs() / (var + eps).sqrt()).median()), float((mu.abs() / (var + eps).sqrt()).max())
    print(f"{tag:64s} |mu|/sigma median {ratio[0]:6.3f} max {ratio[1]:6.2f} | max err / max|out|: LN then GEMM {ea:.2e}  folded {eb:.2e} | rms: {ra:.2e}  {rb:.2e}", flush=True)


torch.manual_seed(0)
seen = set()
for tag, x, g, b, eps, blk, which in captured:
    if tag in seen:
        continue
    seen.add(tag)
    W, bias = consumer_weight(blk, which)
    compare(tag, x, g, b, eps, W, bias)
print("-- the same rows with their mean inflated (a trained residual stream carries large per-token offsets in a few channels / in the mean):")
tag, x, g, b, eps, blk, which = captured[len(captured) // 2]
W, bias = consumer_weight(blk, which)
xf = x.float()
sig = xf.std(-1, keepdim=True)
for k in (3.0, 10.0, 30.0, 100.0):
    compare(f"{tag}  + {k:g} sigma on every channel", (xf + k * sig).to(torch.bfloat16), g, b, eps, W, bias)
print("-- and with trained-looking gamma (log-normal, sigma 0.5) instead of the initial ones:")
g2 = torch.exp(0.5 * torch.randn_like(g))
compare(f"{tag}  gamma ~ logN(0, 0.5)", x, g2, b, eps, W, bias)
compare(f"{tag}  gamma ~ logN(0, 0.5), + 10 sigma", (xf + 10 * sig).to(torch.bfloat16), g2, b, eps, W, bias)
