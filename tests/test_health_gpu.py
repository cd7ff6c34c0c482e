"""Fail-closed handling of a stream-K fix-up that gives up (VERDICT r1 #10 / ADVICE r1): the tile is poisoned, the
process-wide health word is raised, the fused optimizers skip the update on the device and report on the host from the
next call on; nk_health_clear() restores service."""
import json
import os
from functools import partial
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import UNET_TINY, synth_state_dict
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(autouse=True)
def _clean_health():
    from neurosis_amd import lib

    lib.call("nk_health_clear")
    yield
    os.environ.pop("NK_SK_DEBUG", None)
    os.environ.pop("NK_GEMM_G2", None)
    lib.call("nk_health_clear")


def _engine(optimizer):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import DiffusionEngine

    net = D.UNetModel(**UNET_TINY)
    net.load_state_dict(synth_state_dict(json.loads((G / "unet_sdxl_tiny_keys.json").read_text())))
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    eng = DiffusionEngine(model=net, denoiser=den, first_stage_model=None, optimizer=optimizer,
                          loss_fn=D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting())).cuda()
    eng.setup_flat_params()
    return eng


@pytest.mark.parametrize("opt", ["adafactor", "adamw"])
def test_flagged_backward_is_not_applied_and_is_reported(opt):
    from neurosis_amd import lib, ops
    from neurosis_amd.optimizers import Adafactor, AdamW

    fx = load_fixture("unet_sdxl_tiny")
    eng = _engine(partial(Adafactor, scale_parameter=True, relative_step=True, warmup_init=True) if opt == "adafactor" else partial(AdamW, lr=1e-3))
    batch = {"crossattn": fx["context"].cuda(), "vector": fx["y"].cuda()}

    def step():
        eng(fx["x"].cuda(), batch, sigmas=fx["sigma"].cuda(), noise=fx["noise"].cuda()).mean().backward()
        eng.optimizer_step()

    step()
    torch.cuda.synchronize()
    m0, s0 = eng.store.master.clone(), eng.store.shadow.clone()
    lib.call("nk_debug_raise_health", ops._stream())          # what a give-up inside this step's backward does
    step()                                                      # the update kernels see the word and touch nothing
    torch.cuda.synchronize()
    assert torch.equal(eng.store.master, m0) and torch.equal(eng.store.shadow, s0)
    assert lib.query("nk_health_status") == 1
    with pytest.raises(lib.NkError, match="health"):            # ... and the next update refuses on the host
        step()
    lib.call("nk_health_clear")
    assert lib.query("nk_health_status") == 0
    step()
    torch.cuda.synchronize()
    assert not torch.equal(eng.store.master, m0)


def test_stream_k_give_up_poisons_the_tile_and_raises_the_word():
    from neurosis_amd import lib, ops

    os.environ["NK_GEMM_G2"] = "0"                              # (this shape would otherwise take the two-group kernel, which has no K split)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, 5120, generator=g).bfloat16().cuda()
    w = (torch.randn(1280, 5120, generator=g) * 0.02).bfloat16().cuda()
    ref = ops.gemm_nt(x, w)                                     # 320 tiles x 80 k-steps: the persistent stream-K kernel, K-split tiles
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ref.float()).all()) and lib.query("nk_health_status") == 0
    os.environ["NK_SK_DEBUG"] = "2"                             # every fix-up wait gives up at once
    bad = ops.gemm_nt(x, w)
    torch.cuda.synchronize()
    os.environ.pop("NK_SK_DEBUG")
    assert bool(torch.isnan(bad.float()).any()), "a tile whose partials were not joined must not look like a result"
    assert lib.query("nk_health_status") == 1 and lib.query("nk_gemm_sk_status") == 1
    lib.call("nk_health_clear")
    again = ops.gemm_nt(x, w)
    torch.cuda.synchronize()
    assert torch.equal(again, ref) and lib.query("nk_health_status") == 0
