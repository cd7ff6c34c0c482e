"""The oracle's restatement of the loss CLASS (oracle.sdxl_oracle.diffusion_loss / noise_with_offset / training_step_loss)
against fixtures produced by the reference's own `StandardDiffusionLoss._forward`, `apply_noise_offset` and
`DiffusionEngine.training_step / encode_first_stage` (tests/golden/make_golden.py: loss_class_case, engine_case), and this
package's host-side `apply_noise_offset` against the reference's under the same seed.  CPU only."""
import json
from pathlib import Path

import pytest
import torch

from oracle import sdxl_oracle as O
from tests.golden.make_golden import LOSS_CLASS_CASES, UNET_TINY, VAE_TINY, synth_state_dict
from tests.golden.fixture_io import load_fixture

G = Path(__file__).parent / "golden"
TOL = 1e-5


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.fixture(scope="module")
def fx():
    return load_fixture("loss_class_tiny")


@pytest.fixture(scope="module")
def unet_sd():
    return {k: v.clone().requires_grad_(True) for k, v in synth_state_dict(json.loads((G / "unet_sdxl_tiny_keys.json").read_text())).items()}


@pytest.mark.parametrize("tag", [t for t, _ in LOSS_CLASS_CASES])
def test_oracle_loss_class_matches_reference(fx, unet_sd, tag):
    case = fx["cases"][tag]
    kw = case["kwargs"]
    for v in unet_sd.values():
        v.grad = None
    noise = O.noise_with_offset(case["noise"], case["offset"], kw.get("noise_offset", 0.0))
    net = lambda xin, t: O.unet_forward(unet_sd, UNET_TINY, xin, t, fx["context"], fx["y"])
    loss = O.diffusion_loss(net, O.legacy_ddpm_sigmas(), fx["x"], case["sigma"], noise, kw["loss_type"], kw["objective_type"])
    assert rel(loss.detach(), case["loss"]) <= TOL, (loss.tolist(), case["loss"].tolist())
    loss.mean().backward()
    for k, g in case["grads"].items():
        assert rel(unet_sd[k].grad, g) <= 5e-4, k
    gmax = max(case["grad_norms"].values())
    for k, n in case["grad_norms"].items():   # (biases in front of a GroupNorm have a mathematically zero gradient: rounding noise)
        assert abs(float(unet_sd[k].grad.norm()) - n) <= 1e-3 * max(n, 1e-4 * gmax), k


def test_apply_noise_offset_matches_reference_under_the_same_seed(fx):
    """neurosis_amd's DiffusionLoss.apply_noise_offset is host logic: same draw, same arithmetic as loss.py:32-40"""
    import neurosis_amd.modules.diffusion as D

    for tag, kw in LOSS_CLASS_CASES:
        case = fx["cases"][tag]
        lf = D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting(), **kw)
        torch.manual_seed(case["offset_seed"])
        got = lf.apply_noise_offset(case["noise"].clone(), fx["x"])
        assert torch.equal(got, case["offset_out"]), tag
    with_offset = fx["cases"]["edm_l2_offset"]
    assert not torch.equal(with_offset["offset_out"], with_offset["noise"])       # the offset branch really ran in the reference
    # clamping of the constructor arguments (loss.py:27-30)
    lf = D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting(), noise_offset=3.0, noise_offset_chance=-1.0)
    assert lf.noise_offset == 1.0 and lf.noise_offset_chance == 0.0


def test_oracle_training_step_matches_reference_engine():
    """DiffusionEngine.encode_first_stage (chunked by vae_batch_size) and training_step of the reference itself"""
    e = load_fixture("engine_tiny")
    keys = json.loads((G / "engine_tiny_keys.json").read_text())
    usd = synth_state_dict(keys["unet"])
    vsd_all = synth_state_dict(keys["vae"])
    vsd = {k[len("encoder."):]: v for k, v in vsd_all.items() if k.startswith("encoder.")}
    vsd.update({k: v for k, v in vsd_all.items() if k.startswith("quant_conv.")})
    dd = {k: v for k, v in VAE_TINY.items() if k not in ("embed_dim", "standalone")}
    mean, per, latents = O.training_step_loss(usd, UNET_TINY, vsd, dd, e["scale_factor"], e["image"], e["sigma"], e["noise"], e["crossattn"], e["vector"])
    assert rel(latents, e["latents"]) <= TOL
    assert abs(float(mean) - float(e["loss_mean"])) <= TOL * abs(float(e["loss_mean"]))
    assert abs(float(per[0]) - float(e["logged"]["train/loss_s0"])) <= TOL * abs(float(per[0]))
    assert abs(float(mean) - float(e["logged"]["train/loss"])) <= TOL * abs(float(mean))


def test_engine_state_dict_keys_equal_the_reference_engines():
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import AutoencoderKL, DiffusionEngine

    keys = json.loads((G / "engine_tiny_keys.json").read_text())
    eng = DiffusionEngine(model=D.UNetModel(**UNET_TINY), denoiser=D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000,
                          discretization=D.LegacyDDPMDiscretization()), first_stage_model=AutoencoderKL(embed_dim=4, ddconfig={k: v for k, v in VAE_TINY.items() if k != "embed_dim"}),
                          loss_fn=D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting()), scale_factor=0.13025, input_key="image")
    assert set(eng.state_dict().keys()) == set(keys["engine_state_dict_keys"])
