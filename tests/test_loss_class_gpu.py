"""HIP path against fixtures produced by the reference's own `StandardDiffusionLoss._forward / get_loss /
apply_noise_offset` (edm/l2, edm/l1, edm/l2 + noise offset, rf/l2) and its own `DiffusionEngine.encode_first_stage /
training_step` (tests/golden/make_golden.py: loss_class_case, engine_case).  Tolerances as in test_modules_gpu.py
(bf16 MFMA operands vs an fp32 CPU reference): per-sample loss 1e-2 relative, gradient cosine >= 0.9985 (l1: 0.995, rf: 0.998, through the VAE: 0.996), latents 3e-2."""
import json
from functools import partial
from pathlib import Path
from unittest import mock

import pytest
import torch

from tests.golden.make_golden import LOSS_CLASS_CASES, UNET_TINY, VAE_TINY, synth_state_dict
from tests.util import check_grad_cosines, cosine, rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _unet(store=True):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.nn import FlatParamStore

    net = D.UNetModel(**UNET_TINY)
    net.load_state_dict(synth_state_dict(json.loads((G / "unet_sdxl_tiny_keys.json").read_text())))
    net = net.cuda()
    return net, (FlatParamStore(net.parameters()) if store else None)


@pytest.mark.parametrize("tag", [t for t, _ in LOSS_CLASS_CASES])
def test_hip_loss_class_matches_the_reference_class(tag):
    import neurosis_amd.modules.diffusion as D

    fx = load_fixture("loss_class_tiny")
    case = fx["cases"][tag]
    kw = case["kwargs"]
    net, st = _unet()
    if kw["objective_type"] == "rf":
        den, weighting = D.Denoiser(preconditioning=D.RectifiedFlowXLPreconditioning()), D.RectifiedFlowWeighting()
    else:
        den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization()).cuda()
        weighting = D.EpsWeighting()
    assert type(weighting).__name__ == case["weighting"]
    lossfn = D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=weighting, **kw)
    cond = {"crossattn": fx["context"].cuda(), "vector": fx["y"].cuda()}
    # the per-(sample, channel) offset the reference drew: apply_noise_offset asks torch.randn for it (loss.py:37)
    with mock.patch("torch.randn", side_effect=lambda *a, **k: case["offset"].clone()) if case["offset"] is not None else mock.patch("builtins.id", id):
        loss = lossfn._forward(D.OpenAIWrapper(net), den, cond, fx["x"].cuda(), {}, sigmas=case["sigma"].cuda(), noise=case["noise"].cuda())
    assert loss.shape == case["loss"].shape and loss.dtype == torch.float32
    assert rel_err(loss.detach(), case["loss"]) <= 1e-2, (loss.tolist(), case["loss"].tolist())
    loss.mean().backward()
    torch.cuda.synchronize()
    named = dict(net.named_parameters())
    gmax = max(case["grad_norms"].values())
    # measured worst >= 2-D / 1-D: edm_l2 0.99909 / 0.99936, edm_l2_offset 0.99918 / 0.99947, rf_l2 0.99896 / 0.99928, edm_l1 0.99620 / 0.99768 (the
    # l1 loss's gradient is a SIGN: one flipped element of the bf16 network output moves it by a whole unit)
    fm, fv = {"edm_l1": (0.995, 0.995), "rf_l2": (0.998, 0.998)}.get(tag, (0.9985, 0.998))
    check_grad_cosines(f"loss class {tag}", named, case["grads"], floor_matrix=fm, floor_vector=fv, keep=lambda k, g: float(g.norm()) > 1e-2 * gmax)
    for k, n in case["grad_norms"].items():
        if n > 1e-2 * gmax:
            assert abs(float(named[k].grad.norm()) - n) <= 6e-2 * n, (tag, k)


def _engine(**kw):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import AutoencoderKL, DiffusionEngine

    keys = json.loads((G / "engine_tiny_keys.json").read_text())
    net = D.UNetModel(**UNET_TINY)
    net.load_state_dict(synth_state_dict(keys["unet"]))
    vae = AutoencoderKL(embed_dim=4, ddconfig={k: v for k, v in VAE_TINY.items() if k != "embed_dim"})
    # (ddconfig.standalone=true gives the reference's Encoder / Decoder their own quant convs, which _init_first_stage then
    # overwrites with the autoencoder's top-level ones, models/diffusion.py:157-162: those four entries carry no information)
    vae.load_state_dict({k: v for k, v in synth_state_dict(keys["vae"]).items() if not k.startswith(("encoder.quant_conv", "decoder.post_quant_conv"))})
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    eng = DiffusionEngine(model=net, denoiser=den, first_stage_model=vae, scale_factor=0.13025, input_key="image", vae_batch_size=2,
                          loss_fn=D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting()), **kw).cuda()
    eng.setup_flat_params()
    return eng


def test_engine_methods_match_the_reference_engine():
    e = load_fixture("engine_tiny")
    eng = _engine()
    batch = {"image": e["image"].cuda(), "crossattn": e["crossattn"].cuda(), "vector": e["vector"].cuda()}
    latents = eng.encode_first_stage(eng.get_input(batch))          # 3 images in chunks of vae_batch_size = 2
    assert latents.shape == e["latents"].shape
    assert rel_err(latents.float(), e["latents"]) <= 3e-2 and cosine(latents.float(), e["latents"]) >= 0.999
    loss = eng.training_step(batch, 0, sigmas=e["sigma"].cuda(), noise=e["noise"].cuda())
    assert abs(float(loss) - float(e["loss_mean"])) <= 1e-2 * float(e["loss_mean"])
    for k, v in e["logged"].items():                                   # the reference's log_dict entries (models/diffusion.py:228-231)
        assert abs(float(eng.last_log[k]) - float(v)) <= 1.5e-2 * abs(float(v)), k
    loss.backward()
    torch.cuda.synchronize()
    named = dict(eng.model.diffusion_model.named_parameters())
    # (through the bf16 VAE encoder as well: measured worst 0.99702 / 0.99708)
    check_grad_cosines("reference engine training_step", named, e["grads"], floor_matrix=0.996, floor_vector=0.996, keep=lambda k, g: float(g.norm()) > 1e-6)


def test_conditioner_beside_the_vae_encoder_gives_the_in_line_loss(monkeypatch):
    """training_step runs the frozen conditioner on a side stream beside the frozen VAE encoder (NK_COND_OVERLAP, default on) and joins it before
    the UNet: same kernels on the same data -- the loss and every gradient equal the in-line order's bit for bit, step after step."""
    e = load_fixture("engine_tiny")
    batch = lambda: {"image": e["image"].cuda(), "crossattn": e["crossattn"].cuda(), "vector": e["vector"].cuda()}
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NK_COND_OVERLAP", mode)
        eng = _engine()
        losses = []
        for _ in range(3):
            loss = eng.training_step(batch(), 0, sigmas=e["sigma"].cuda(), noise=e["noise"].cuda())
            loss.backward()
            torch.cuda.synchronize()
            losses.append(loss.detach().clone())
        res[mode] = (losses, eng.store.grad.clone())
        assert (getattr(eng, "_cond_stream", None) is not None) == (mode == "1")
    assert all(torch.equal(a, b) for a, b in zip(res["0"][0], res["1"][0])) and torch.equal(res["0"][1], res["1"][1])


def test_optimizer_and_ema_state_survive_a_checkpoint_round_trip():
    """Adafactor's factored second moments / step count and the EMA shadow are training state.  Two steps, save
    (engine.state_dict + optimizer.state_dict), load into a fresh engine, third step on both: bitwise equal parameters,
    optimizer state and EMA.  The EMA entries carry LitEma's names (modules/ema.py:23-29)."""
    from neurosis_amd.optimizers import Adafactor, AdafactorScheduler

    e = load_fixture("engine_tiny")
    mk = lambda: _engine(optimizer=partial(Adafactor, scale_parameter=True, relative_step=True, warmup_init=True),
                         scheduler=partial(AdafactorScheduler, initial_lr=4e-7), use_ema=True, ema_decay_rate=0.99)
    batch = lambda: {"image": e["image"].cuda(), "crossattn": e["crossattn"].cuda(), "vector": e["vector"].cuda()}

    def step(eng, seed):
        g = torch.Generator(device="cuda").manual_seed(seed)
        noise = torch.randn(e["noise"].shape, device="cuda", generator=g)
        eng.training_step(batch(), 0, sigmas=e["sigma"].cuda(), noise=noise).backward()
        eng.optimizer_step()

    a = mk()
    assert isinstance(a._torch_optimizer, Adafactor) and a.adafactor is a._torch_optimizer.flat
    step(a, 1)
    step(a, 2)
    sd = {k: v.detach().clone() for k, v in a.state_dict().items()}
    osd = a._torch_optimizer.state_dict()
    ema_keys = [k for k in sd if k.startswith("model_ema.")]
    assert "model_ema.decay" in sd and "model_ema.num_updates" in sd and int(sd["model_ema.num_updates"]) == 2
    assert "model_ema.diffusion_modelinput_blocks00weight" in sd and sd["model_ema.diffusion_modelinput_blocks00weight"].shape == (32, 4, 3, 3)
    assert len(ema_keys) == 2 + len(a.store.params)
    st0 = osd["state"][0]
    assert int(st0["step"]) == 2 and "RMS" in st0 and ("exp_avg_sq_row" in st0 or "exp_avg_sq" in st0)

    b = mk()
    missing, unexpected = b.load_state_dict(sd, strict=True)
    b.store.refresh()
    b._torch_optimizer.load_state_dict(osd)
    assert b.adafactor.step_count == 2
    assert torch.equal(b.model_ema.shadow, a.model_ema.shadow) and torch.equal(b.store.master, a.store.master)
    assert torch.equal(b.adafactor.state, a.adafactor.state)
    # a control that resumes WITHOUT the optimizer state (what round 1 did): step count 0 -> relative step 1e-6 * 1 instead of 1e-6 * 3
    c = mk()
    c.load_state_dict(sd, strict=True)
    c.store.refresh()
    before = a.store.master.double().clone()
    step(a, 3)
    step(b, 3)
    step(c, 3)
    torch.cuda.synchronize()
    # (bitwise equality is not on offer: the tiny-grid weight gradients sum their K splits with fp32 atomics)
    upd_a, upd_b, upd_c = (x.store.master.double() - before for x in (a, b, c))
    assert float((upd_a - upd_b).norm() / upd_a.norm()) <= 1e-3
    assert float((upd_a - upd_c).norm() / upd_a.norm()) >= 0.3, "resuming without the optimizer state must be visibly different"
    assert float((a.adafactor.state - b.adafactor.state).norm() / a.adafactor.state.norm()) <= 1e-4
    assert float((a.model_ema.shadow - b.model_ema.shadow).abs().max()) <= 1e-6
    assert a.adafactor.step_count == b.adafactor.step_count == 3 and c.adafactor.step_count == 1
    lr_a, lr_b = a._torch_scheduler.get_last_lr()[0], b._torch_scheduler.get_last_lr()[0]
    assert abs(lr_a - lr_b) <= 1e-6 * lr_a and lr_a != 4e-7


def test_masters_changed_refreshes_derived_state():
    """a checkpoint load / broadcast after the optimizer exists must invalidate Adafactor's per-tile sums of
    p^2 and, before the first update, re-seed the EMA from the new weights."""
    from neurosis_amd.optimizers import Adafactor

    eng = _engine(optimizer=partial(Adafactor, scale_parameter=True, relative_step=True, warmup_init=True), use_ema=True)
    eng.adafactor.refresh_param_norms()
    assert eng.adafactor._p2_valid
    with torch.no_grad():
        eng.store.master.mul_(2.0)
    eng.store.masters_changed()
    assert not eng.adafactor._p2_valid
    assert torch.equal(eng.model_ema.shadow, eng.store.master)
    assert torch.equal(eng.store.shadow.float(), eng.store.master.bfloat16().float())


def test_lightning_adapter_step_logic_with_a_stand_in_trainer():
    """neurosis_amd.trainer.DiffusionEngineMI355X without Lightning installed: manual optimization, accumulate_grad_batches
    handled by DiffusionEngine.accumulate (first micro-batch overwrites, the last one steps), optimizer state in the checkpoint."""
    from types import SimpleNamespace

    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import AutoencoderKL
    from neurosis_amd.optimizers import Adafactor
    from neurosis_amd.trainer import DiffusionEngineMI355X

    e = load_fixture("engine_tiny")
    keys = json.loads((G / "engine_tiny_keys.json").read_text())
    net = D.UNetModel(**UNET_TINY)
    net.load_state_dict(synth_state_dict(keys["unet"]))
    vae = AutoencoderKL(embed_dim=4, ddconfig={k: v for k, v in VAE_TINY.items() if k != "embed_dim"})
    vae.load_state_dict({k: v for k, v in synth_state_dict(keys["vae"]).items() if not k.startswith(("encoder.quant_conv", "decoder.post_quant_conv"))})
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    ad = DiffusionEngineMI355X(require_lightning=False, model=net, denoiser=den, first_stage_model=vae, scale_factor=0.13025, input_key="image",
                               optimizer=partial(Adafactor, scale_parameter=True, relative_step=True, warmup_init=True),
                               loss_fn=D.StandardDiffusionLoss(sigma_generator=D.EDMSigmaGenerator(), loss_weighting=D.EpsWeighting()))
    ad._trainer_stub = SimpleNamespace(device=torch.device("cuda", 0), world_size=1, accumulate_grad_batches=2)
    ad.on_fit_start()
    eng = ad.engine
    batch = lambda: {"image": e["image"].cuda(), "crossattn": e["crossattn"].cuda(), "vector": e["vector"].cuda()}
    m0 = eng.store.master.clone()
    l0 = ad.training_step(batch(), 0)
    torch.cuda.synchronize()
    g0 = eng.store.grad.clone()
    assert torch.equal(eng.store.master, m0) and eng.global_step == 0 and eng.store.state.grad_accumulate is False
    l1 = ad.training_step(batch(), 1)                     # second micro-batch: adds, then the fused update
    eng.join_optimizer()
    torch.cuda.synchronize()
    assert eng.global_step == 1 and not torch.equal(eng.store.master, m0)
    assert float(eng.store.grad.norm()) > 0 and not torch.equal(eng.store.grad, g0)
    assert torch.isfinite(l0) and torch.isfinite(l1) and not l0.requires_grad
    ckpt = {}
    ad.on_save_checkpoint(ckpt)
    assert ckpt["nk_global_step"] == 1 and int(ckpt["nk_optimizer"]["state"][0]["step"]) == 1


def test_lightning_adapter_applies_a_checkpoint_that_arrives_before_fit_start():
    """Trainer.fit(ckpt_path=...) calls load_state_dict and on_load_checkpoint BEFORE on_fit_start, i.e. before the flat
    store, the fused optimizer and the EMA module exist.  The adapter parks what arrives early and applies it at the end of
    on_fit_start: the resumed adapter continues with the saved step count, second moments and EMA, and a strict load succeeds."""
    from types import SimpleNamespace

    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models import AutoencoderKL
    from neurosis_amd.optimizers import Adafactor
    from neurosis_amd.trainer import DiffusionEngineMI355X

    e = load_fixture("engine_tiny")
    keys = json.loads((G / "engine_tiny_keys.json").read_text())

    def mk():
        net = D.UNetModel(**UNET_TINY)
        net.load_state_dict(synth_state_dict(keys["unet"]))
        vae = AutoencoderKL(embed_dim=4, ddconfig={k: v for k, v in VAE_TINY.items() if k != "embed_dim"})
        vae.load_state_dict({k: v for k, v in synth_state_dict(keys["vae"]).items() if not k.startswith(("encoder.quant_conv", "decoder.post_quant_conv"))})
        den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
        ad = DiffusionEngineMI355X(require_lightning=False, model=net, denoiser=den, first_stage_model=vae, scale_factor=0.13025, input_key="image",
                                   use_ema=True, ema_decay_rate=0.99, optimizer=partial(Adafactor, scale_parameter=True, relative_step=True, warmup_init=True),
                                   loss_fn=D.StandardDiffusionLoss(sigma_generator=D.EDMSigmaGenerator(), loss_weighting=D.EpsWeighting()))
        ad._trainer_stub = SimpleNamespace(device=torch.device("cuda", 0), world_size=1, accumulate_grad_batches=1)
        return ad

    batch = lambda: {"image": e["image"].cuda(), "crossattn": e["crossattn"].cuda(), "vector": e["vector"].cuda()}
    a = mk()
    a.on_fit_start()
    torch.manual_seed(5)
    a.training_step(batch(), 0)
    a.training_step(batch(), 1)
    a.engine.join_optimizer()
    torch.cuda.synchronize()
    ckpt = {"state_dict": {k: v.detach().clone() for k, v in a.state_dict().items()}}
    a.on_save_checkpoint(ckpt)
    assert any(k.startswith("engine.model_ema.") for k in ckpt["state_dict"])

    b = mk()                                               # Lightning's order: restore first, on_fit_start afterwards
    assert b.engine.store is None and b.engine._torch_optimizer is None
    b.load_state_dict(ckpt["state_dict"], strict=True)
    b.on_load_checkpoint(ckpt)
    assert b._pending_optimizer is not None and b._pending_ema is not None
    b.on_fit_start()
    assert b._pending_optimizer is None and b._pending_ema is None
    assert b.engine.global_step == 2 and b.engine.adafactor.step_count == 2
    assert torch.equal(b.engine.store.master, a.engine.store.master)
    assert torch.equal(b.engine.adafactor.state, a.engine.adafactor.state)
    assert torch.equal(b.engine.model_ema.shadow, a.engine.model_ema.shadow) and int(b.engine.model_ema.num_updates) == 2
