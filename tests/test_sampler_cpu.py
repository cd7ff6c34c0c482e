"""Sampling loop, CPU side: the oracle restatement and the product's sampler classes (pure host logic over
torch ops when no FusedDenoiser is involved) against trajectories captured from the reference's own sampler classes
(tests/golden/make_golden.py::sampler_cases).  fp32 vs fp32: tolerance 2e-6 relative."""
from pathlib import Path

import json
import pytest
import torch

from oracle import sampler_oracle as SO
from oracle import sdxl_oracle as O
from tests.golden.make_golden import SAMPLER_CASES, analytic_denoiser, sampler_inputs, seeded_noise_sampler, synth_state_dict
from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"
TOL = 2e-6


def _oracle_run(name):
    cls, kw, scale, steps = SAMPLER_CASES[name]
    x0, cond, uc = sampler_inputs()
    sigmas = O.legacy_ddpm_sampling_sigmas(steps)
    denoise = SO.cfg_denoise(analytic_denoiser, scale, cond, uc)
    torch.manual_seed(4321)
    if cls == "EulerEDMSampler":
        return SO.edm(denoise, x0, sigmas, heun=False, **kw)
    if cls == "HeunEDMSampler":
        return SO.edm(denoise, x0, sigmas, heun=True, **kw)
    if cls == "EulerAncestralSampler":
        return SO.euler_ancestral(denoise, x0, sigmas, seeded_noise_sampler(77), **kw)
    if cls == "DPMPP2SAncestralSampler":
        return SO.euler_ancestral(denoise, x0, sigmas, seeded_noise_sampler(77), dpmpp2s=True, **kw)
    if cls == "DPMPP2MSampler":
        return SO.dpmpp2m(denoise, x0, sigmas)
    if cls == "LinearMultistepSampler":
        return SO.lms(denoise, x0, sigmas, **kw)
    raise KeyError(cls)


@pytest.mark.parametrize("name", sorted(SAMPLER_CASES))
def test_oracle_samplers_match_reference(name):
    want = load_fixture("sampler_analytic")[name]
    assert rel_err(_oracle_run(name), want) < TOL


def product_sampler(name, device="cpu"):
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd.modules.diffusion import LegacyDDPMDiscretization
    from neurosis_amd.modules.guidance import VanillaCFG

    cls, kw, scale, steps = SAMPLER_CASES[name]
    sampler = getattr(S, cls)(discretization=LegacyDDPMDiscretization(), guider=None if scale is None else VanillaCFG(scale), num_steps=steps,
                              device=device, **kw)
    if hasattr(sampler, "noise_sampler"):
        sampler.noise_sampler = seeded_noise_sampler(77)
    return sampler


@pytest.mark.parametrize("name", sorted(SAMPLER_CASES))
def test_product_samplers_match_reference_on_cpu(name):
    want = load_fixture("sampler_analytic")[name]
    x0, cond, uc = sampler_inputs()
    torch.manual_seed(4321)
    with torch.no_grad():
        got = product_sampler(name)(analytic_denoiser, x0.clone(), cond, uc=uc)
    assert rel_err(got, want) < TOL


def test_sampling_sigma_table_matches_product_discretization():
    from neurosis_amd.modules.diffusion import LegacyDDPMDiscretization

    for n in (4, 7, 50, 999, 1000):
        assert torch.equal(O.legacy_ddpm_sampling_sigmas(n), LegacyDDPMDiscretization()(n))


def test_oracle_unet_sampling_trajectories():
    """Euler+CFG, Heun+CFG and plain Euler through the oracle UNet and the oracle eps-denoiser against the reference's
    trajectories (tiny SDXL-style UNet, 16x16 latents)."""
    fx = load_fixture("sampler_unet_tiny")
    shapes = json.loads((G / "unet_sdxl_tiny_keys.json").read_text())
    sd = synth_state_dict(shapes)
    cfg = load_fixture("unet_sdxl_tiny")["cfg"]
    table = O.legacy_ddpm_sigmas()

    def denoiser(x, sigma, c):
        return O.eps_denoiser(lambda xin, t: O.unet_forward(sd, cfg, xin, t, c["crossattn"], c["vector"]), table, x, sigma)

    with torch.no_grad():
        for name, run in fx["runs"].items():
            record = []
            final = SO.edm(SO.cfg_denoise(denoiser, run["scale"], fx["cond"], fx["uc"]), fx["noise"].clone(), O.legacy_ddpm_sampling_sigmas(run["steps"]),
                           heun=run["cls"] == "HeunEDMSampler", record=record)
            assert len(record) == len(run["trajectory"])
            for got, want in zip(record, run["trajectory"]):
                assert rel_err(got, want) < 2e-5, name
            assert rel_err(final, run["final"]) < 2e-5, name


def test_oracle_vae_decoder():
    fx = load_fixture("vae_decoder_tiny")
    sd = synth_state_dict(json.loads((G / "vae_decoder_tiny_keys.json").read_text()))
    assert rel_err(O.vae_decode(sd, fx["cfg"], fx["z"]), fx["image"]) < 1e-5
