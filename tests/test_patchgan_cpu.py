"""oracle/patchgan_oracle.py and the product's discriminator losses against the reference's NLayerDiscriminator / loss functions
(tests/golden/make_golden.py::discriminator_case).  fp32 vs fp32: 1e-5."""
import json
from pathlib import Path

import pytest
import torch

from oracle import patchgan_oracle as PO
from tests.golden.make_golden import disc_state_dict
from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"
FX = load_fixture("patchgan_tiny")
SHAPES = json.loads((G / "patchgan_tiny_keys.json").read_text())


@pytest.mark.parametrize("kind", ["hinge", "vanilla"])
def test_discriminator_step_oracle(kind):
    case = FX["cases"][kind]
    sd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in disc_state_dict(SHAPES).items()}
    running = {}
    real = PO.discriminator(sd, FX["real"], running=running)
    fake = PO.discriminator(sd, FX["fake"], running=running)
    assert rel_err(real, case["logits_real"]) < 1e-5 and rel_err(fake, case["logits_fake"]) < 1e-5
    loss = PO.disc_loss(kind, real, fake)
    assert abs(float(loss) - float(case["d_loss"])) <= 1e-6 * abs(float(case["d_loss"])) + 1e-7
    loss.backward()
    for k, g in case["grads"].items():
        assert rel_err(sd[k].grad, g) < 1e-4, k
    for k, v in running.items():
        assert rel_err(v, case["buffers"][k]) < 1e-5, k


def test_generator_term_oracle():
    sd = disc_state_dict(SHAPES)
    img = FX["fake"].clone().requires_grad_(True)
    g_loss = -PO.discriminator(sd, img).mean()
    g_loss.backward()
    assert abs(float(g_loss) - float(FX["generator"]["g_loss"])) <= 1e-6 * abs(float(FX["generator"]["g_loss"])) + 1e-7
    assert rel_err(img.grad, FX["generator"]["d_image"]) < 1e-4


@pytest.mark.parametrize("kind", ["hinge", "vanilla"])
def test_product_disc_losses_and_their_gradients(kind):
    from neurosis_amd.modules.losses import get_discr_loss_fn

    case = FX["cases"][kind]
    real, fake = (case[k].clone().requires_grad_(True) for k in ("logits_real", "logits_fake"))
    PO.disc_loss(kind, real, fake).backward()
    loss, d_real, d_fake = get_discr_loss_fn(kind).with_grad(case["logits_real"], case["logits_fake"])
    assert abs(float(loss) - float(case["d_loss"])) <= 1e-6
    assert rel_err(d_real, real.grad) < 1e-6 and rel_err(d_fake, fake.grad) < 1e-6
    late = get_discr_loss_fn(kind, start_step=100)
    assert float(late(case["logits_real"], case["logits_fake"], global_step=5)) == 0.0


def test_generator_adversarial_pieces_oracle_vs_reference():
    """oracle/patchgan_oracle.generator_adversarial_loss against what the reference's own GeneralLPIPSWithDiscriminator.forward computes for
    the generator branch when it is given a tensor `weights` (tests/golden/make_golden.py::gan_generator_case): nll, g, adaptive weight."""
    from tests.golden.make_golden import synth_state_dict

    gfx = load_fixture("gan_generator_tiny")
    G = Path(__file__).resolve().parent / "golden"
    vfx = load_fixture("vae_train_tiny")
    sd = synth_state_dict(json.loads((G / "vae_train_tiny_keys.json").read_text()))
    dsd = disc_state_dict(json.loads((G / "patchgan_tiny_keys.json").read_text()))
    enc = {k[len("encoder."):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
    dec = {k[len("decoder."):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    hp = gfx["hp"]
    loss, nll, g, dw, xrec = PO.generator_adversarial_loss(enc, dec, dsd, vfx["cfg"], gfx["x"], gfx["noise"], rec_weight=hp["rec_weight"], logvar=hp["logvar_init"],
                                                           disc_factor=hp["disc_factor"], disc_weight=hp["disc_weight"])
    assert rel_err(xrec, gfx["xrec"]) <= 1e-5
    assert abs(float(nll) - float(gfx["nll"])) <= 1e-5 * abs(float(gfx["nll"]))
    assert abs(float(g) - float(gfx["g_loss"])) <= 1e-4 * abs(float(gfx["g_loss"])) + 1e-7
    assert abs(float(dw) - float(gfx["d_weight"])) <= 1e-3 * float(gfx["d_weight"])
