"""PatchGAN discriminator on the GPU: BatchNorm / LeakyReLU kernels against their formulas, the discriminator's
training-mode forward + backward (weight gradients, running statistics, input gradient) against the reference's
NLayerDiscriminator (tests/golden/patchgan_tiny.pt).

Tolerances: kernels 2e-2 (bf16 in/out, fp32 statistics); network: logits 3e-2 of max magnitude / cosine 0.999, losses 1e-2,
parameter gradients cosine >= 0.985 (measured 0.988-1.000; the two layers nearest the input, behind three BatchNorm backward
passes in bf16, are the low end) and norms within 6e-2 for the convolution weights, cosine >= 0.97 / norms within 8e-2 for the
8-64-element BatchNorm and bias vectors, running statistics 1e-2.  The backward is driven by d loss / d logits evaluated on the
REFERENCE's logits: the hinge loss's masks (real < 1, fake > -1) are step functions, and a logit that lands on the other side
of the threshold in bf16 changes the summed gradients discretely -- that is the loss's discontinuity, not the network's error.
LeakyReLU has the same kink per activation, which is why the input gradient is compared in the L2 sense.
"""
import json
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import disc_state_dict
from tests.util import cosine, rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("slope", [1.0, 0.2])
@pytest.mark.parametrize("M,C", [(4 * 32 * 32, 16), (777, 64), (36, 8)])
def test_batchnorm_kernels(M, C, slope):
    from neurosis_amd import ops

    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, C, generator=g) * 1.7 + 0.4).to(torch.bfloat16).cuda()
    w, b = (torch.randn(C, generator=g).cuda() for _ in range(2))
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    dy = torch.randn(M, C, generator=g).to(torch.bfloat16).cuda()
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = torch.nn.functional.batch_norm(xr, rm_ref, rv_ref, wr, br, training=True, momentum=0.1, eps=1e-5)
    ref = torch.nn.functional.leaky_relu(ref, slope) if slope != 1.0 else ref
    ref.backward(dy.float())
    w.grad = b.grad = None
    y, bwd = ops.batchnorm_fwd(x, w, b, rm, rv, 1e-5, 0.1, slope)
    assert rel_err(y, ref) <= 2e-2
    assert rel_err(rm, rm_ref) <= 1e-3 and rel_err(rv, rv_ref) <= 1e-3
    dx = bwd(dy)
    # the reference differentiates through fp32 activations; where the bf16 output rounds to the other side of 0 the LeakyReLU
    # branch differs: compare on the bulk
    assert cosine(dx, xr.grad) >= 0.995 and rel_err(dx, xr.grad) <= 8e-2
    assert rel_err(w.grad, wr.grad) <= 3e-2 and rel_err(b.grad, br.grad) <= 3e-2


def test_leaky_relu_kernels():
    from neurosis_amd import ops

    x = torch.linspace(-4, 4, 4096).to(torch.bfloat16).cuda()
    y, bwd = ops.leaky_relu_fwd(x, 0.2)
    want = torch.nn.functional.leaky_relu(x.float(), 0.2)
    assert rel_err(y, want) <= 4e-3
    dy = torch.ones_like(x)
    assert torch.equal(bwd(dy).float(), torch.where(x.float() >= 0, 1.0, 0.2).to(torch.bfloat16).float())


def _disc():
    from neurosis_amd.modules.losses import NLayerDiscriminator

    fx = load_fixture("patchgan_tiny")
    shapes = json.loads((G / "patchgan_tiny_keys.json").read_text())
    disc = NLayerDiscriminator(**fx["cfg"])
    disc.load_state_dict(disc_state_dict(shapes), strict=False)
    return fx, disc.cuda().train()


def _tokens(img):
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    B, C, H, W = img.shape
    return Img(ops.nchw_to_tokens(img.cuda().float().contiguous(), 8), B, H, W)


@pytest.mark.parametrize("kind", ["hinge", "vanilla"])
def test_discriminator_step_against_reference(kind):
    from neurosis_amd import ops
    from neurosis_amd.modules.losses import get_discr_loss_fn

    fx, disc = _disc()
    case = fx["cases"][kind]
    lr, b_real = disc.fwdb(_tokens(fx["real"]), need_dx=False)
    lf, b_fake = disc.fwdb(_tokens(fx["fake"]), need_dx=False)
    B = fx["real"].shape[0]
    real = ops.tokens_to_nchw(lr.t, B, 1, lr.H, lr.W, dtype=torch.float32)
    fake = ops.tokens_to_nchw(lf.t, B, 1, lf.H, lf.W, dtype=torch.float32)
    for got, want in ((real, case["logits_real"]), (fake, case["logits_fake"])):
        assert got.shape == want.shape and rel_err(got, want) <= 3e-2 and cosine(got, want) >= 0.999
    loss, _, _ = get_discr_loss_fn(kind).with_grad(real, fake)
    assert abs(float(loss) - float(case["d_loss"])) <= 1e-2 * float(case["d_loss"])
    _, d_real, d_fake = get_discr_loss_fn(kind).with_grad(case["logits_real"].cuda(), case["logits_fake"].cuda())
    dstate = ops.state_of(next(disc.parameters()))
    dstate.grad_accumulate = False
    b_real(ops.nchw_to_tokens(d_real.contiguous(), 8))
    dstate.grad_accumulate = True                         # the second pass adds to the first one's weight gradients
    try:
        b_fake(ops.nchw_to_tokens(d_fake.contiguous(), 8))
    finally:
        dstate.grad_accumulate = False
    ops.join_wgrad_stream()
    torch.cuda.synchronize()
    grads = dict(disc.named_parameters())
    report = {k: (round(cosine(grads[k].grad, g), 4), round(float(grads[k].grad.norm()) / max(float(g.norm()), 1e-12), 4)) for k, g in case["grads"].items()}
    bad = {k: v for k, v in report.items()
           if v[0] < (0.97 if case["grads"][k].dim() == 1 else 0.985) or abs(v[1] - 1.0) > (8e-2 if case["grads"][k].dim() == 1 else 6e-2)}
    assert not bad, (bad, report)
    state = disc.state_dict()
    for k, v in case["buffers"].items():
        if "num_batches" in k:
            assert int(state[k]) == int(v)
        else:
            assert rel_err(state[k], v) <= 1e-2, k


def test_generator_term_input_gradient():
    from neurosis_amd import ops

    fx, disc = _disc()
    logits, bwd = disc.fwdb(_tokens(fx["fake"]))
    B = fx["fake"].shape[0]
    vals = ops.tokens_to_nchw(logits.t, B, 1, logits.H, logits.W, dtype=torch.float32)
    g_loss = -vals.mean()
    assert abs(float(g_loss) - float(fx["generator"]["g_loss"])) <= 1e-2 * abs(float(fx["generator"]["g_loss"])) + 1e-3
    d_logits = torch.full_like(vals, -1.0 / vals.numel())
    d_img = bwd(ops.nchw_to_tokens(d_logits.contiguous(), 8))
    got = ops.tokens_to_nchw(d_img, B, 3, 64, 64, dtype=torch.float32)
    want = fx["generator"]["d_image"]
    assert cosine(got, want) >= 0.99 and float((got.cpu() - want).norm() / want.norm()) <= 0.13
    assert disc(fx["fake"].cuda()).shape == (4, 1, 6, 6)


def test_eval_mode_uses_running_statistics():
    from oracle import patchgan_oracle as PO  # noqa: F401  (same formula as torch's eval-mode batch_norm below)

    fx, disc = _disc()
    disc.eval()
    got = disc(fx["real"].cuda())
    shapes = json.loads((G / "patchgan_tiny_keys.json").read_text())
    sd = disc_state_dict(shapes)
    h = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(fx["real"], sd["layers.0.weight"], sd["layers.0.bias"], stride=2, padding=1), 0.2)
    for idx, stride in ((2, 2), (5, 2), (8, 1)):
        h = torch.nn.functional.conv2d(h, sd[f"layers.{idx}.weight"], None, stride=stride, padding=1)
        bn = f"layers.{idx + 1}"
        h = torch.nn.functional.batch_norm(h, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"], training=False)
        h = torch.nn.functional.leaky_relu(h, 0.2)
    want = torch.nn.functional.conv2d(h, sd["layers.11.weight"], sd["layers.11.bias"], stride=1, padding=1)
    assert got.shape == want.shape and rel_err(got, want) <= 3e-2 and cosine(got, want) >= 0.999
    assert int(disc.layers[3].num_batches_tracked) == 0
