"""LPIPS on the GPU: max-pool / ReLU / layer-distance kernels against their formulas, and the product's
LPIPS (AlexNet / VGG16 trunk on the tile engine, fused layer distance, backward to the second image) against the reference's LPIPS.forward
(tests/golden/lpips_{vgg,alex}_tiny.pt).  Tolerances: kernels 1e-2; distance 5e-3 relative (measured 9e-4); image gradient cosine >= 0.98 and relative
L2 error <= 0.2 (measured 0.988 / 0.15: 13 bf16 convolutions and 13 ReLU kinks deep, gradient magnitudes ~1e-5)."""
import json
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

from tests.golden.make_golden import synth_state_dict
from tests.util import cosine, rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def test_maxpool_and_relu_kernels():
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    g = torch.Generator().manual_seed(2)
    N, H, W, C = 2, 12, 10, 24
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16)
    xr = x.float().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous().cuda()
    y, bwd = ops.maxpool2x2_fwd(Img(tok(x), N, H, W))
    assert torch.equal(y.t.float().cpu(), tok(ref.detach().to(torch.bfloat16)).float().cpu())
    assert torch.equal(bwd(tok(dy)).float().cpu(), tok(xr.grad.to(torch.bfloat16)).float().cpu())
    # AlexNet's overlapping 3x3 / 2 pool (odd and even extents; values quantised so that ties occur and the first-maximum rule matters)
    for Hh, Ww in ((15, 11), (8, 6)):
        xq = (torch.randn(N, C, Hh, Ww, generator=g) * 2).round().div(2).to(torch.bfloat16)
        xr = xq.float().requires_grad_(True)
        ref = F.max_pool2d(xr, 3, 2)
        dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
        ref.backward(dy.float())
        y, bwd = ops.maxpool_fwd(Img(tok(xq), N, Hh, Ww), 3, 2)
        assert (y.H, y.W) == tuple(ref.shape[2:])
        assert torch.equal(y.t.float().cpu(), tok(ref.detach().to(torch.bfloat16)).float().cpu())
        assert rel_err(bwd(tok(dy)).float().cpu(), tok(xr.grad).float().cpu()) <= 4e-3      # sums of up to four bf16 gradients, rounded once
    v = torch.linspace(-3, 3, 1024).to(torch.bfloat16).cuda()
    r, b = ops.leaky_relu_fwd(v, 0.0)
    assert torch.equal(r, torch.relu(v)) and torch.equal(b(torch.ones_like(v)).float(), (v > 0).float())


@pytest.mark.parametrize("C,HW", [(64, 100), (512, 16), (200, 7)])
def test_lpips_layer_kernel(C, HW):
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    g = torch.Generator().manual_seed(3)
    N = 3
    f0, f1 = (torch.randn(N, HW, C, generator=g).abs().to(torch.bfloat16) for _ in range(2))
    w = torch.rand(C, generator=g)
    up = torch.tensor([1.0, 0.5, 2.0])
    b = f1.float().requires_grad_(True)
    unit = lambda t: t / (t.pow(2).sum(-1, keepdim=True).sqrt() + 1e-10)
    ref = ((unit(f0.float()) - unit(b)).pow(2) * w).sum(-1).mean(-1)
    (ref * up).sum().backward()
    out = torch.full((N,), 7.0, device="cuda")
    bwd = ops.lpips_layer(Img(f0.reshape(-1, C).cuda(), N, HW, 1), Img(f1.reshape(-1, C).cuda(), N, HW, 1), w.cuda(), out, accumulate=False)
    assert rel_err(out, ref) <= 1e-4
    got = bwd(up.cuda()).float().reshape(N, HW, C)
    assert rel_err(got, b.grad) <= 1e-2
    ops.lpips_layer(Img(f0.reshape(-1, C).cuda(), N, HW, 1), Img(f1.reshape(-1, C).cuda(), N, HW, 1), w.cuda(), out, accumulate=True)
    assert rel_err(out, 2 * ref) <= 1e-4


def _lpips(kind="vgg"):
    from neurosis_amd.modules.losses import LPIPS

    fx = load_fixture(f"lpips_{kind}_tiny")
    shapes = json.loads((G / f"lpips_{kind}_tiny_keys.json").read_text())
    lp = LPIPS(pnet_type=kind, lin_weights=fx["lin"])
    lp.load_state_dict({k: v * 1.6 for k, v in synth_state_dict(shapes).items()}, strict=False)
    for k, v in fx["lin"].items():
        assert torch.equal(lp.state_dict()[k], v)
    assert not any(p.requires_grad for p in lp.parameters())
    return fx, lp.cuda()


@pytest.mark.parametrize("kind", ["vgg", "alex"])
def test_lpips_against_reference(kind):
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    fx, lp = _lpips(kind)
    dist = lp(fx["x"].cuda(), fx["y"].cuda())
    assert dist.shape == fx["distance"].shape and rel_err(dist, fx["distance"]) <= 5e-3, (dist.reshape(-1).tolist(), fx["distance"].reshape(-1).tolist())
    B, C, H, W = fx["y"].shape
    out, bwd = lp.fwdb(fx["x"].cuda(), Img(ops.nchw_to_tokens(fx["y"].cuda().contiguous(), 8), B, H, W))
    d_tok = bwd(fx["upstream"].cuda())
    got = ops.tokens_to_nchw(d_tok, B, 3, H, W, dtype=torch.float32)
    want = fx["d_y"]
    assert cosine(got, want) >= 0.98, cosine(got, want)
    assert float((got.cpu() - want).norm() / want.norm()) <= 0.2
    assert float(d_tok[:, 3:].float().abs().max()) == 0.0


def test_lpips_needs_weights_or_says_so():
    from neurosis_amd.modules.losses import LPIPS

    assert LPIPS(pretrained=False).pnet_type == "alex"            # the reference's default trunk (perceptual.py:67)
    with pytest.raises(KeyError):
        LPIPS(pnet_type="squeeze", pretrained=False)
    try:
        import neurosis.data  # noqa: F401
    except Exception:
        with pytest.raises(RuntimeError, match="lin weights"):
            LPIPS(pnet_type="vgg")
