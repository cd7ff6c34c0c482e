"""Autoencoder reconstruction training step on the GPU (reconstruction part): AutoencodingEngine
(Encoder.fwdb -> DiagonalGaussianRegularizer -> Decoder.fwdb -> fused l2 loss -> explicit backward) against the reference's
Encoder / regularizer / Decoder stack on the same weights, image and posterior noise (tests/golden/vae_train_tiny.pt), plus
the pieces it adds (softmax backward, unfused attention backward).

Tolerances as for the UNet (bf16 activations vs fp32 CPU): outputs 3e-2 of max magnitude / cosine 0.999, loss 1e-2 relative,
parameter gradients cosine >= 0.99, gradient norms within 5e-2 relative (+ an absolute floor for analytically-zero ones).
"""
import json
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import synth_state_dict
from tests.util import cosine, rel_err

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _engine(loss="l2", **kw):
    from neurosis_amd.models.autoencoder import AutoencodingEngine, DiagonalGaussianRegularizer
    from neurosis_amd.modules.diffusion.model import Decoder, Encoder

    fx = torch.load(G / "vae_train_tiny.pt", weights_only=False)
    sd = synth_state_dict(json.loads((G / "vae_train_tiny_keys.json").read_text()))
    eng = AutoencodingEngine(encoder=Encoder(**fx["cfg"]), decoder=Decoder(**fx["cfg"]), loss=loss, regularizer=DiagonalGaussianRegularizer(sample=True), **kw)
    eng.load_state_dict(sd)
    eng = eng.cuda()
    eng.setup_flat_params()
    return fx, eng


def test_softmax_rows_backward_kernel():
    from neurosis_amd.lib import call

    g = torch.Generator().manual_seed(0)
    M, L, scale = 96, 200, 0.37
    logits = torch.randn(M, L, generator=g)
    p = logits.softmax(-1).to(torch.bfloat16).cuda()
    dp = torch.randn(M, L, generator=g).to(torch.bfloat16).cuda()
    pf, df = p.float(), dp.float()
    want = pf * (df - (df * pf).sum(-1, keepdim=True)) * scale
    call("nk_softmax_rows_bwd", p.data_ptr(), dp.data_ptr(), M, L, scale, torch.cuda.current_stream().cuda_stream)
    assert rel_err(dp, want) <= 1e-2


def test_unfused_attention_backward_vs_autograd():
    from neurosis_amd import ops

    g = torch.Generator().manual_seed(1)
    B, L, D = 2, 64, 128
    q, k, v, do = (torch.randn(B * L, D, generator=g).to(torch.bfloat16) for _ in range(4))
    qr, kr, vr = (t.float().reshape(B, L, D).requires_grad_(True) for t in (q, k, v))
    ref = ((qr @ kr.transpose(1, 2)) * D ** -0.5).softmax(-1) @ vr
    ref.backward(do.float().reshape(B, L, D))
    o, bwd = ops.attention_unfused_fwd(q.cuda(), k.cuda(), v.cuda(), B)
    assert rel_err(o, ref.reshape(B * L, D)) <= 2e-2
    for got, want in zip(bwd(do.cuda()), (qr.grad, kr.grad, vr.grad)):
        assert rel_err(got, want.reshape(B * L, D)) <= 3e-2 and cosine(got, want.reshape(B * L, D)) >= 0.999


@pytest.mark.parametrize("tag", ["rec_only", "rec_kl"])
def test_reconstruction_step_against_reference(tag):
    fx0 = torch.load(G / "vae_train_tiny.pt", weights_only=False)
    case = fx0["cases"][tag]
    fx, eng = _engine(regularization_weights={"kl_loss": case["kl_weight"]} if case["kl_weight"] else None)
    loss, z, xrec, reg_log = eng.loss_and_backward(fx["x"].cuda(), noise=case["noise"].cuda())
    assert rel_err(z, case["z"]) <= 3e-2 and cosine(z, case["z"]) >= 0.999
    assert rel_err(xrec, case["xrec"]) <= 3e-2 and cosine(xrec, case["xrec"]) >= 0.999
    assert abs(float(loss) - float(case["loss"])) <= 1e-2 * abs(float(case["loss"]))
    assert abs(float(reg_log["kl_loss"]) - float(case["kl_loss"])) <= 1e-2 * float(case["kl_loss"])
    grads = dict(eng.named_parameters())
    for k, g in case["grads"].items():
        c = cosine(grads[k].grad, g)
        assert c >= 0.99, (k, c)
    bad = []
    gmax = max(case["grad_norms"].values())
    for k, n in case["grad_norms"].items():
        mine = float(grads[k].grad.float().norm())
        if abs(mine - n) > 5e-2 * n + 2e-3 * gmax:
            bad.append((k, mine, n))
    assert not bad, bad[:8]


def test_training_steps_reduce_the_loss_and_match_eval_forward():
    fx, eng = _engine()
    x = fx["x"].cuda()
    noise = fx["cases"]["rec_only"]["noise"].cuda()
    first = float(eng.training_step({"image": x}, 0, lr=2e-3, noise=noise))
    for i in range(1, 8):
        last = float(eng.training_step({"image": x}, i, lr=2e-3, noise=noise))
    assert last < 0.8 * first, (first, last)
    # the forward-only path (what DiffusionEngine uses) sees the updated weights and agrees with the training forward
    eng.regularization.sample = False
    z, xrec, _ = eng(x)
    _, z2, xrec2, _ = eng.loss_and_backward(x)
    assert rel_err(z, z2) <= 1e-2 and rel_err(xrec, xrec2) <= 1e-2


def test_l1_loss_branch_and_gan_loss_is_refused():
    from neurosis_amd.models.autoencoder import AutoencodingEngine

    fx, eng = _engine(loss="l1")
    loss, _, xrec, _ = eng.loss_and_backward(fx["x"].cuda(), noise=fx["cases"]["rec_only"]["noise"].cuda())
    assert abs(float(loss) - float((xrec - fx["x"].cuda()).abs().mean())) <= 1e-6
    assert float(eng.store.grad.abs().max()) > 0
    with pytest.raises(NotImplementedError):
        AutoencodingEngine(encoder=eng.encoder, decoder=eng.decoder, loss=torch.nn.Identity())
