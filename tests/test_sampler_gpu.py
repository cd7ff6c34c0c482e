"""Sampling path on the GPU: the nk_sample_* kernels against their formulas, every sampler class on
device tensors against the reference's trajectories, the fused UNet sampling path (FusedDenoiser + EulerEDM / Heun + CFG)
against trajectories the reference produced with the same tiny UNet, and the VAE decoder against the reference's output.

Tolerances: kernels -- prepare is bit-exact (one RNE rounding of an fp32 product), denoise / euler 2e-6 (fp32, FMA
contraction); sampler classes on an analytic denoiser 5e-6 (fp32 on both sides); UNet trajectories 4e-2 of the latent's max
magnitude with cosine >= 0.999 (bf16 network, error compounding over the steps); decoder 3e-2 / cosine 0.999 as the encoder.
"""
import json
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import SAMPLER_CASES, analytic_denoiser, sampler_inputs, synth_state_dict
from tests.test_sampler_cpu import product_sampler
from tests.util import cosine, rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _stream():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("rep", [1, 2])
@pytest.mark.parametrize("shape", [(3, 4, 6, 5), (2, 4, 128, 128), (1, 9, 7, 3)])
def test_sample_kernels_against_formulas(rep, shape):
    from neurosis_amd.lib import call

    B, C, H, W = shape
    HW, cpad = H * W, (C + 7) // 8 * 8
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).cuda()
    c_in, c_skip, c_out = (torch.rand(B, generator=g).cuda() + 0.1 for _ in range(3))
    sigma_hat = torch.rand(B, generator=g).cuda() * 10 + 0.5
    sigma_next = sigma_hat * 0.7
    scale = 5.5

    net_in = torch.full((rep * B * HW, cpad), 7.0, dtype=torch.bfloat16, device="cuda")
    call("nk_sample_prepare", x.data_ptr(), c_in.data_ptr(), net_in.data_ptr(), B, C, HW, cpad, rep, _stream())
    want = (x * c_in[:, None, None, None]).permute(0, 2, 3, 1).reshape(B, HW, C).to(torch.bfloat16)
    got = net_in.reshape(rep, B, HW, cpad)
    for r in range(rep):
        assert torch.equal(got[r, :, :, :C], want)
        assert float(got[r, :, :, C:].float().abs().max()) == 0.0 if cpad > C else True

    net_out = torch.randn(rep * B * HW, cpad, generator=g).to(torch.bfloat16).cuda()
    f = net_out.float().reshape(rep, B, H, W, cpad)[..., :C].permute(0, 1, 4, 2, 3)
    guided = f[0] if rep == 1 else f[0] + scale * (f[1] - f[0])
    want_d = c_skip[:, None, None, None] * x + c_out[:, None, None, None] * guided
    den = torch.empty_like(x)
    call("nk_sample_denoise", net_out.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), scale, den.data_ptr(), B, C, HW, cpad, rep, _stream())
    assert rel_err(den, want_d) <= 2e-6

    want_x = x + (sigma_next - sigma_hat)[:, None, None, None] * ((x - want_d) / sigma_hat[:, None, None, None])
    x_next, den2 = torch.empty_like(x), torch.empty_like(x)
    call("nk_sample_euler_step", net_out.data_ptr(), x.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), sigma_hat.data_ptr(), sigma_next.data_ptr(),
         scale, x_next.data_ptr(), den2.data_ptr(), B, C, HW, cpad, rep, _stream())
    assert rel_err(x_next, want_x) <= 2e-6
    assert torch.equal(den2, den)
    # in place, without the optional output
    x_inplace = x.clone()
    call("nk_sample_euler_step", net_out.data_ptr(), x_inplace.data_ptr(), c_skip.data_ptr(), c_out.data_ptr(), sigma_hat.data_ptr(),
         sigma_next.data_ptr(), scale, x_inplace.data_ptr(), None, B, C, HW, cpad, rep, _stream())
    assert torch.equal(x_inplace, x_next)


def test_sample_kernels_reject_bad_shapes():
    from neurosis_amd.lib import NkError, call

    x = torch.zeros(1, 4, 2, 2, device="cuda")
    v = torch.ones(1, device="cuda")
    buf = torch.zeros(64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(NkError):
        call("nk_sample_prepare", x.data_ptr(), v.data_ptr(), buf.data_ptr(), 1, 4, 4, 4, 1, _stream())      # Cpad not a multiple of 8
    with pytest.raises(NkError):
        call("nk_sample_prepare", x.data_ptr(), v.data_ptr(), buf.data_ptr(), 1, 4, 4, 8, 3, _stream())      # rep must be 1 or 2


@pytest.mark.parametrize("name", sorted(n for n in SAMPLER_CASES if "churn" not in n))
def test_sampler_classes_on_device_match_reference(name):
    """(the churn case draws torch.randn_like from the device generator and is pinned on the CPU only)"""
    want = load_fixture("sampler_analytic")[name]
    x0, cond, uc = sampler_inputs()
    cond, uc = ({k: v.cuda() for k, v in d.items()} for d in (cond, uc))
    with torch.no_grad():
        got = product_sampler(name, "cuda")(analytic_denoiser, x0.cuda(), cond, uc=uc)
    assert got.is_cuda and rel_err(got, want) <= 5e-6


def _tiny_engine(sampler):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.models.diffusion import DiffusionEngine

    shapes = json.loads((G / "unet_sdxl_tiny_keys.json").read_text())
    cfg = load_fixture("unet_sdxl_tiny")["cfg"]
    net = D.UNetModel(**cfg)
    net.load_state_dict(synth_state_dict(shapes))
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    return DiffusionEngine(net, den, None, sampler=sampler).cuda().eval()


@pytest.mark.parametrize("run", ["euler_cfg", "heun_cfg", "euler_plain"])
def test_fused_unet_sampling_against_reference_trajectory(run):
    import neurosis_amd.modules.diffusion as D
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd.modules.guidance import VanillaCFG

    fx = load_fixture("sampler_unet_tiny")
    ref = fx["runs"][run]
    sampler = getattr(S, ref["cls"])(discretization=D.LegacyDDPMDiscretization(), guider=None if ref["scale"] is None else VanillaCFG(ref["scale"]),
                                     num_steps=ref["steps"])
    engine = _tiny_engine(sampler)
    cond, uc = ({k: v.cuda() for k, v in d.items()} for d in (fx["cond"], fx["uc"]))
    trajectory = []
    step = sampler.sampler_step
    sampler.sampler_step = lambda *a, **k: trajectory.append(step(*a, **k).clone()) or trajectory[-1]   # (clone: graph replays reuse one buffer)
    fused_calls = []
    originals = {kernel: getattr(S.FusedDenoiser, kernel) for kernel in ("euler", "guided")}

    def spy_on(kernel):
        def spy(self, *a, **k):
            fused_calls.append(kernel)
            return originals[kernel](self, *a, **k)
        return spy

    for kernel in originals:
        setattr(S.FusedDenoiser, kernel, spy_on(kernel))
    try:
        final = engine.sample(cond, uc=uc, batch_size=2, shape=(4, 16, 16), noise=fx["noise"])
    finally:
        for kernel, fn in originals.items():
            setattr(S.FusedDenoiser, kernel, fn)
    # the HIP route ran: Euler = one fused step per interval; Heun = two guided evaluations per interval but the last
    expect = ["euler"] * ref["steps"] if ref["cls"] == "EulerEDMSampler" else ["guided"] * (2 * ref["steps"] - 1)
    assert fused_calls == expect
    assert len(trajectory) == len(ref["trajectory"])
    for got, want in zip(trajectory, ref["trajectory"]):
        assert rel_err(got, want) <= 4e-2, (run, rel_err(got, want))
        assert cosine(got, want) >= 0.999
    assert rel_err(final, ref["final"]) <= 4e-2


def test_fused_route_equals_generic_route_on_the_same_network():
    """the same bf16 UNet behind both routes: what differs is only where the elementwise work runs (kernels vs torch ops on
    fp32 NCHW, one extra bf16 rounding of the network output on the generic route's NHWC->NCHW fp32 conversion: none)."""
    import neurosis_amd.modules.diffusion as D
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd.modules.guidance import VanillaCFG

    fx = load_fixture("sampler_unet_tiny")
    sampler = S.EulerEDMSampler(discretization=D.LegacyDDPMDiscretization(), guider=VanillaCFG(5.0), num_steps=4)
    engine = _tiny_engine(sampler)
    cond, uc = ({k: v.cuda() for k, v in d.items()} for d in (fx["cond"], fx["uc"]))
    fused = engine.sample(cond, uc=uc, batch_size=2, shape=(4, 16, 16), noise=fx["noise"])

    def generic_cb(inputs, sigma, c):
        return engine.denoiser(engine.model, inputs, sigma, c, "D")

    with torch.no_grad():
        generic = sampler(generic_cb, fx["noise"].cuda().clone(), cond, uc=uc)
    assert rel_err(fused, generic) <= 1e-2 and cosine(fused, generic) >= 0.9999


def test_captured_step_equals_eager_step_and_survives_weight_updates():
    """The hipGraph of a sampling step replays the same kernels on the same buffers: same latents as launching them one by
    one.  A second sample() with other conditioning reuses the graph (conditioning is copied in); after the parameters change
    the step is captured again and sees the new weights."""
    import neurosis_amd.modules.diffusion as D
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd import ops
    from neurosis_amd.modules.guidance import VanillaCFG

    fx = load_fixture("sampler_unet_tiny")
    sampler = S.EulerEDMSampler(discretization=D.LegacyDDPMDiscretization(), guider=VanillaCFG(5.0), num_steps=4)
    engine = _tiny_engine(sampler)
    cond, uc = ({k: v.cuda() for k, v in d.items()} for d in (fx["cond"], fx["uc"]))
    eager = S.FusedDenoiser(engine.model, engine.denoiser, use_graph=False)
    graphed = S.FusedDenoiser(engine.model, engine.denoiser, use_graph=True)
    with torch.no_grad():
        want = sampler(eager, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        got = sampler(graphed, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        assert rel_err(got, want) <= 1e-6
        assert len(graphed._captured) == 1
        graph = next(iter(graphed._captured.values()))
        cond2 = {k: v.flip(0).contiguous() for k, v in cond.items()}
        want2 = sampler(eager, fx["noise"].cuda().clone(), cond2, uc=uc).clone()
        got2 = sampler(graphed, fx["noise"].cuda().clone(), cond2, uc=uc).clone()
        assert next(iter(graphed._captured.values())) is graph
        assert rel_err(got2, want2) <= 1e-6 and rel_err(got2, want) > 1e-3
        for p in engine.model.parameters():
            p.mul_(1.01)
        ops.state.param_epoch += 1
        want3 = sampler(eager, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        got3 = sampler(graphed, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        assert next(iter(graphed._captured.values())) is not graph
        assert rel_err(got3, want3) <= 1e-6 and rel_err(got3, want) > 1e-4


def test_captured_step_sees_a_store_refresh_of_the_channel_padded_convs():
    """with the parameters in a flat store the graph's key does not change when the masters do; the channel-padded stand-ins
    of the 4-channel convolutions (conv_in, out) must therefore be refilled INSIDE the graph.  Replaying after store.refresh() with
    changed weights equals the eager step on the new weights (it mixed old conv_in / out weights with new ones before)."""
    import neurosis_amd.modules.diffusion as D
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd.modules.guidance import VanillaCFG
    from neurosis_amd.nn import FlatParamStore

    fx = load_fixture("sampler_unet_tiny")
    sampler = S.EulerEDMSampler(discretization=D.LegacyDDPMDiscretization(), guider=VanillaCFG(5.0), num_steps=3)
    engine = _tiny_engine(sampler)
    store = FlatParamStore(engine.model.parameters())
    cond, uc = ({k: v.cuda() for k, v in d.items()} for d in (fx["cond"], fx["uc"]))
    eager = S.FusedDenoiser(engine.model, engine.denoiser, use_graph=False)
    graphed = S.FusedDenoiser(engine.model, engine.denoiser, use_graph=True)
    with torch.no_grad():
        first = sampler(graphed, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        graph = next(iter(graphed._captured.values()))
        store.master.mul_(1.05)                      # what an optimizer step or ema_scope()'s copy_to does: masters move, then refresh
        store.refresh()
        want = sampler(eager, fx["noise"].cuda().clone(), cond, uc=uc).clone()
        got = sampler(graphed, fx["noise"].cuda().clone(), cond, uc=uc).clone()
    assert next(iter(graphed._captured.values())) is graph, "store-managed parameters keep the captured graph"
    assert rel_err(got, want) <= 1e-6 and rel_err(got, first) > 1e-4


def test_vae_decoder_against_reference_golden():
    import neurosis_amd.modules.diffusion as D

    fx = load_fixture("vae_decoder_tiny")
    dec = D.Decoder(**fx["cfg"])
    dec.load_state_dict(synth_state_dict(json.loads((G / "vae_decoder_tiny_keys.json").read_text())))
    dec = dec.cuda()
    image = dec(fx["z"].cuda())
    assert image.shape == fx["image"].shape and image.dtype == torch.float32
    assert rel_err(image, fx["image"]) <= 3e-2, rel_err(image, fx["image"])
    assert cosine(image, fx["image"]) >= 0.999
    dec.max_batch_size = 1
    chunks = dec(fx["z"].cuda())
    assert isinstance(chunks, list) and len(chunks) == 2
    assert torch.equal(torch.cat(chunks, 0), dec(fx["z"].cuda(), cat_zero=True))


def test_engine_decode_and_log_images_round_trip():
    """AutoencoderKL wiring through the engine: encode_first_stage / decode_first_stage share the scale factor, log_images
    returns inputs, reconstructions and CFG samples of the right shapes."""
    import neurosis_amd.modules.diffusion as D
    import neurosis_amd.modules.diffusion.sampling as S
    from neurosis_amd.models.autoencoder import AutoencoderKL
    from neurosis_amd.models.diffusion import DiffusionEngine
    from neurosis_amd.modules.guidance import VanillaCFG

    shapes = json.loads((G / "unet_sdxl_tiny_keys.json").read_text())
    cfg = load_fixture("unet_sdxl_tiny")["cfg"]
    net = D.UNetModel(**cfg)
    net.load_state_dict(synth_state_dict(shapes))
    dd = dict(load_fixture("vae_decoder_tiny")["cfg"])
    dd.pop("embed_dim")
    vae = AutoencoderKL(embed_dim=4, ddconfig=dd)
    enc_sd = synth_state_dict(json.loads((G / "vae_encoder_tiny_keys.json").read_text()))
    dec_sd = synth_state_dict(json.loads((G / "vae_decoder_tiny_keys.json").read_text()))
    state = {f"encoder.{k}": v for k, v in enc_sd.items() if not k.startswith("quant_conv")}
    state.update({f"decoder.{k}": v for k, v in dec_sd.items() if not k.startswith("post_quant_conv")})
    state.update({k: v for k, v in enc_sd.items() if k.startswith("quant_conv")})
    state.update({k: v for k, v in dec_sd.items() if k.startswith("post_quant_conv")})
    vae.load_state_dict(state)
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    sampler = S.EulerEDMSampler(discretization=D.LegacyDDPMDiscretization(), guider=VanillaCFG(3.0), num_steps=3)
    engine = DiffusionEngine(net, den, vae, sampler=sampler, scale_factor=0.13025, input_key="image").cuda().eval()
    g = torch.Generator().manual_seed(5)
    batch = {"image": (torch.rand(2, 3, 128, 128, generator=g) * 2 - 1).cuda(), "crossattn": torch.randn(2, 7, cfg["context_dim"], generator=g).cuda(),
             "vector": torch.randn(2, cfg["adm_in_channels"], generator=g).cuda()}
    z = engine.encode_first_stage(batch["image"])
    fx_dec = load_fixture("vae_decoder_tiny")
    direct = engine.vae_decoder(fx_dec["z"].cuda())
    assert rel_err(engine.decode_first_stage(fx_dec["z"].cuda() * 0.13025), direct) <= 1e-6
    out = engine.log_images(batch, num_img=2)
    assert out["train/inputs"].shape == (2, 3, 128, 128) and out["train/recons"].shape == (2, 3, 128, 128)
    assert out["samples"].shape == (2, 3, 128, 128) and torch.isfinite(out["samples"]).all()
    assert z.shape == (2, 4, 16, 16)
