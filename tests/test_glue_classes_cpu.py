"""Every preconditioning / weighting / sigma-generator / discretisation class of SURVEY rows A4-A5 against values captured
from the reference's classes on fixed inputs (tests/golden/make_golden.py::glue_class_cases).  Host-side [B]-sized fp32/fp64
arithmetic on both sides: 2e-6 relative (tables and pure formulas are exact)."""
from pathlib import Path

import pytest
import torch

import neurosis_amd.modules.diffusion as D
from tests.golden.make_golden import GLUE_CLASSES, GLUE_SIGMAS, GLUE_T
from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

FX = load_fixture("glue_classes")
SIG = torch.tensor(GLUE_SIGMAS)
COMFY = torch.tensor([0.001, 0.1, 0.37, 0.5, 0.93, 0.999])
T = torch.tensor(GLUE_T, dtype=torch.float64)


def _same(got, want, tol=2e-6):
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    finite = torch.isfinite(want)
    assert torch.equal(got[~finite], want[~finite])          # t = 0 gives logSNR = inf, sigma = 0 on both sides
    assert rel_err(got[finite], want[finite]) <= tol, rel_err(got[finite], want[finite])


@pytest.mark.parametrize("name", sorted(GLUE_CLASSES["preconditioning"]))
def test_preconditioning(name):
    obj = getattr(D, name.split("/")[0])(**GLUE_CLASSES["preconditioning"][name])
    for got, want in zip(obj(SIG), FX["preconditioning"][name]):
        _same(got, want)
    for i, getter in enumerate(("get_c_skip", "get_c_out", "get_c_in", "get_c_noise")):
        _same(getattr(obj, getter)(SIG), FX["preconditioning"][name][i])


@pytest.mark.parametrize("name", sorted(FX["weighting"]))
def test_weighting(name):
    if name.startswith("MinSNRGamma"):
        obj = D.MinSNRGammaModifier(D.EpsWeighting(), gamma=5) if name.endswith("eps") else D.MinSNRGammaModifier(D.EDMWeighting(0.5), gamma=3, v_pred=True)
        _same(obj(SIG), FX["weighting"][name])
        return
    obj = getattr(D, name.split("/")[0])(**GLUE_CLASSES["weighting"][name])
    _same(obj(COMFY if "Comfy" in name else SIG), FX["weighting"][name])


@pytest.mark.parametrize("name", sorted(GLUE_CLASSES["generator"]))
def test_sigma_generators(name):
    gen = getattr(D, name.split("/")[0])(**GLUE_CLASSES["generator"][name])
    _same(gen(len(T), T.float() if "Cosine" in name else T), FX["generator"][name])


def test_cosine_generator_shift_and_logsnr():
    _same(D.CosineScheduleSigmaGenerator()(len(T), T.float(), shift=2, return_logSNR=True), FX["generator"]["CosineScheduleSigmaGenerator/shift"])


@pytest.mark.parametrize("name", sorted(GLUE_CLASSES["discretization"]))
def test_discretizations(name):
    disc = getattr(D, name.split("/")[0])(**GLUE_CLASSES["discretization"][name])
    want = FX["discretization"][name]
    for got, ref in zip((disc(1000), disc(10), disc(10, flip=True)), want):
        _same(got, ref, tol=1e-6)


def test_rf_objective_generic_route_on_cpu_matches_reference_formula():
    """StandardDiffusionLoss(objective_type="rf") through the generic (unfused) route with a stand-in network: the loss the
    reference's formula gives (z_t = (1 - sigma) x + sigma eps; mse(F, eps) * w)."""
    class Net(torch.nn.Module):
        def forward(self, x, t, c, **kw):
            return 0.5 * x + t.float().reshape(-1, 1, 1, 1) * 1e-3

    g = torch.Generator().manual_seed(0)
    x, eps = torch.randn(3, 4, 8, 8, generator=g), torch.randn(3, 4, 8, 8, generator=g)
    sigma = torch.tensor([0.2, 0.5, 0.9])
    loss_fn = D.StandardDiffusionLoss(sigma_generator=D.RectifiedFlowComfySigmaGenerator(), loss_weighting=D.RectifiedFlowComfyWeighting(), objective_type="rf")
    den = D.Denoiser(preconditioning=D.RectifiedFlowComfyPreconditioning())
    got = loss_fn._forward(Net(), den, {}, x, {}, sigmas=sigma, noise=eps)
    s = sigma[:, None, None, None]
    z = (1 - s) * x + s * eps
    c_in = (sigma**2 + (1 - sigma) ** 2) ** -0.5
    f = 0.5 * (z * c_in[:, None, None, None]) + (1000.0 * sigma)[:, None, None, None] * 1e-3
    want = ((f - eps) ** 2).flatten(1).mean(1) * D.RectifiedFlowComfyWeighting()(sigma).float()
    assert rel_err(got, want) <= 1e-6
    l1 = D.StandardDiffusionLoss(sigma_generator=D.RectifiedFlowComfySigmaGenerator(), loss_weighting=D.UnitWeighting(), objective_type="rf", loss_type="l1")
    assert rel_err(l1._forward(Net(), den, {}, x, {}, sigmas=sigma, noise=eps), (f - eps).abs().flatten(1).mean(1)) <= 1e-6
    with pytest.raises(ValueError):
        D.StandardDiffusionLoss(sigma_generator=None, loss_weighting=None, objective_type="vp")
