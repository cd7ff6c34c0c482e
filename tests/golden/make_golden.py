#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by running the REFERENCE (neggles/neurosis, read-only at
/root/reference) on the CPU in the authoring container.  The reference never travels to the GPU box; only
the small fixtures written here do.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.safetensors / *.json (no pickle: fixture_io.py)

Third-party packages the reference imports but this image lacks (lightning, torchvision, open_clip, ...) are
replaced by empty stand-in modules at import time only; xformers is left genuinely
absent because the reference handles that ImportError itself.  Nothing of the reference's source is copied:
the fixtures hold inputs, outputs, gradients and the (name, shape) list of its state_dicts.

Weights are a pure function of (parameter name, shape) -- see `synth_state_dict` -- so no checkpoint is
stored; zero-initialised modules get non-zero values so gradients are non-trivial.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import json
import sys
import types
import zlib
from pathlib import Path
from unittest import mock

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests.golden.fixture_io import load_fixture, save_fixture  # noqa: E402

HERE = Path(__file__).resolve().parent
REF_SRC = Path("/root/reference/src")

STUB_ROOTS = {
    "torchvision", "lightning", "lightning_fabric", "pytorch_lightning", "open_clip", "diffusers", "kornia", "jsonargparse", "wandb", "omegaconf",
    "natsort", "pytorch_optimizer", "torchmetrics", "cv2", "pymongo", "adlfs", "pynvml", "bitsandbytes", "deepspeed", "s3fs", "boto3", "botocore",
    "colorcet", "matplotlib", "imageio", "clip", "dreamsim", "lpips", "timm", "peft", "gridfs", "bson", "webdataset", "scipy_stub",
}


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []  # behave as a package
        m.__getattr__ = lambda name: _dummy(spec.name, name)  # type: ignore[attr-defined]
        return m

    def exec_module(self, module):
        if module.__name__ in ("lightning", "lightning.pytorch", "pytorch_lightning"):
            module.__version__ = "2.2.1"       # the reference's pin (pyproject.toml:35); it branches on it (autoencoder.py:61)
            def _freeze(self):          # LightningModule.freeze(): requires_grad False everywhere + eval()
                for p_ in self.parameters():
                    p_.requires_grad = False
                self.eval()

            module.LightningModule = type("LightningModule", (torch.nn.Module,), {"freeze": _freeze})
            module.LightningDataModule = type("LightningDataModule", (), {})
            module.Callback = type("Callback", (), {})


class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _dummy(cls.__name__, name)

    def __call__(cls, *a, **k):
        return type.__call__(cls)

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


def _dummy(mod: str, name: str):
    if name.startswith("__"):
        raise AttributeError(name)
    return _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None, "__class_getitem__": classmethod(lambda cls, item: cls),
                                 "__getattr__": lambda self, n: _dummy(name, n), "__call__": lambda self, *a, **k: self})


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


def import_reference():
    # transformers probes optional packages by importing them: let it look at the real environment before the stubs exist
    import transformers.models.clip.modeling_clip  # noqa: F401
    import transformers.models.clip.tokenization_clip  # noqa: F401
    import transformers.tokenization_utils_base as tub

    # transformers 5.x (this image) moved BatchEncoding out of transformers.tokenization_utils, where the reference
    # (written against 4.36+) imports it from: alias the old location
    import types
    legacy = sys.modules.get("transformers.tokenization_utils")
    if legacy is None or not hasattr(legacy, "BatchEncoding"):
        legacy = legacy or types.ModuleType("transformers.tokenization_utils")
        legacy.BatchEncoding = tub.BatchEncoding
        sys.modules["transformers.tokenization_utils"] = legacy

    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, str(REF_SRC))
    import neurosis.modules.diffusion as nd  # noqa
    import neurosis.modules.diffusion.model as nmodel  # noqa

    return nd, nmodel


# ------------------------------------------------------------------------------------------------
def synth_tensor(name: str, shape, kind_hint: str = "") -> torch.Tensor:
    """Deterministic value for a parameter, a pure function of its name and shape."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    is_norm = any(t in name for t in (".norm", "norm1", "norm2", "norm3", "norm_out", "in_layers.0", "out_layers.0", "out.0", "ln_1", "ln_2", "ln_final",
                                      "layer_norm"))
    if leaf == "weight" and len(shape) == 1 and is_norm:
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if leaf == "bias" or leaf.endswith("_bias"):
        return 0.05 * torch.randn(shape, generator=g)
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) * (1.0 / max(fan_in, 1)) ** 0.5


def synth_state_dict(shapes: dict) -> dict:
    return {k: synth_tensor(k, v) for k, v in shapes.items()}


UNET_TINY = dict(
    in_channels=4, model_channels=32, out_channels=4, num_res_blocks=2, attention_resolutions=[4, 2], channel_mult=[1, 2, 4],
    num_head_channels=16, use_linear_in_transformer=True, transformer_depth=[1, 1, 2], context_dim=64, adm_in_channels=48,
    num_classes="sequential", spatial_transformer_attn_type="torch-sdp", use_checkpoint=False,
)
# SD1.5-style: conv proj_in/out (use_linear False), num_heads instead of head channels, attention at every level, no y
UNET_SD15_TINY = dict(
    in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4],
    num_heads=4, transformer_depth=1, context_dim=48, spatial_transformer_attn_type="torch-sdp", use_checkpoint=False,
)
VAE_TINY = dict(
    ch=32, out_ch=3, ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=64, z_channels=4,
    double_z=True, attn_type="vanilla", embed_dim=4, standalone=True,
)
GRAD_KEYS = [
    "input_blocks.0.0.weight", "out.2.weight", "time_embed.0.weight", "input_blocks.4.1.transformer_blocks.0.attn1.to_q.weight",
    "input_blocks.4.1.transformer_blocks.0.attn2.to_k.weight", "middle_block.1.transformer_blocks.1.ff.net.0.proj.weight",
    "output_blocks.2.2.conv.weight", "input_blocks.3.0.op.weight", "output_blocks.8.0.skip_connection.weight", "middle_block.0.in_layers.0.weight",
    "input_blocks.7.1.norm.bias", "output_blocks.5.1.transformer_blocks.0.norm2.weight", "label_emb.0.2.bias",
]


def unet_case(nd, cfg: dict, name: str, B: int, HW: int, with_y: bool):
    torch.manual_seed(0)
    net = nd.UNetModel(**cfg).eval()
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth_state_dict(shapes))
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 4, HW, HW, generator=g)
    noise = torch.randn(B, 4, HW, HW, generator=g)
    ctx = torch.randn(B, 7, cfg["context_dim"], generator=g)
    y = torch.randn(B, cfg["adm_in_channels"], generator=g) if with_y else None
    sigma = torch.tensor([0.35, 2.7, 9.1, 0.05][:B])

    # the reference's own loss / denoiser stack with injected sigma (fresh denoiser per step)
    denoiser = nd.DiscreteDenoiser(preconditioning=nd.EpsPreconditioning(), num_idx=1000, discretization=nd.LegacyDDPMDiscretization())
    table = denoiser.sigmas.detach().clone()
    wrapper = nd.OpenAIWrapper(net)
    cond = {"crossattn": ctx}
    if with_y:
        cond["vector"] = y
    z_t = x + sigma[:, None, None, None] * noise
    d_out = denoiser(wrapper, z_t, sigma, cond, "D")
    weight = nd.EpsWeighting()(sigma)
    loss = ((d_out.float() - x.float()) ** 2).flatten(1).mean(1) * weight.float()  # BatchMSELoss(mean) * w, loss.py:153-157
    loss.mean().backward()
    idx = denoiser.sigma_to_idx(denoiser.possibly_quantize_sigma(sigma))
    with torch.no_grad():
        f_out = net(z_t * (1.0 / (table[idx] ** 2 + 1.0) ** 0.5)[:, None, None, None], idx, ctx, y)
    grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if k in GRAD_KEYS}
    gnorm = {k: float(p.grad.norm()) for k, p in net.named_parameters()}
    fixture = dict(cfg=cfg, x=x, noise=noise, context=ctx, y=y, sigma=sigma, sigma_table=table, c_noise_idx=idx, z_t=z_t.detach(), F_out=f_out,
                   D_out=d_out.detach(), loss=loss.detach(), grads=grads, grad_norms=gnorm)
    save_fixture(fixture, f"{name}")
    (HERE / f"{name}_keys.json").write_text(json.dumps(shapes, indent=0))
    print(f"{name}: loss={loss.tolist()} params={sum(int(torch.tensor(s).prod()) for s in shapes.values())}")


def vae_case(nmodel):
    enc = nmodel.Encoder(**VAE_TINY).eval()
    shapes = {k: list(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict(synth_state_dict(shapes))
    g = torch.Generator().manual_seed(99)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        z = enc(img, regularize=True)
        moments = enc(img, regularize=False)
    save_fixture(dict(cfg=VAE_TINY, image=img, z=z, moments=moments), "vae_encoder_tiny")
    (HERE / "vae_encoder_tiny_keys.json").write_text(json.dumps(shapes, indent=0))
    print("vae: z", tuple(z.shape), float(z.abs().mean()))


def op_cases(nd):
    """Small known-answer vectors for the scalar glue: timestep embedding and the sigma table."""
    from neurosis.modules.diffusion.util import timestep_embedding

    t = torch.tensor([0, 1, 17, 500, 999])
    emb = timestep_embedding(t, 320)
    disc = nd.LegacyDDPMDiscretization()
    table = disc(1000, do_append_zero=False, flip=False).detach()
    save_fixture(dict(t=t, emb320=emb, ddpm_table=table), "glue_vectors")
    print("glue: table", table.shape, float(table[0]), float(table[-2]), float(table[-1]))


def adafactor_case():
    """Three steps of the reference's Adafactor (optimizers/adafactor.py) on a small mixed parameter set:
    matrices, 3x3 and 1x1 conv weights (OIHW), vectors.  Inputs: initial values and per-step gradients; outputs: the
    parameters after every step and the factored states after the last one."""
    from neurosis.optimizers.adafactor import Adafactor

    g = torch.Generator().manual_seed(1234)
    shapes = [(48, 32), (40, 64), (300, 8), (16, 8, 3, 3), (8, 16, 1, 1), (32,), (7,), (1200,)]
    out = {}
    for tag, kw in [("relative", dict(scale_parameter=True, relative_step=True, warmup_init=True)),
                    ("manual", dict(lr=1e-3, scale_parameter=False, relative_step=False, weight_decay=0.01))]:
        params = [torch.nn.Parameter(torch.randn(*s, generator=g) * (0.5 if len(s) > 1 else 1.0)) for s in shapes]
        init = [p.detach().clone() for p in params]
        opt = Adafactor(params, **kw)
        grads, after = [], []
        for step in range(3):
            gs = [torch.randn(*s, generator=g) * (10.0 ** (step - 1)) for s in shapes]   # small, unit, large: clipping kicks in
            for p, gr in zip(params, gs):
                p.grad = gr.clone()
            opt.step()
            grads.append(gs)
            after.append([p.detach().clone() for p in params])
        states = []
        for p in params:
            st = opt.state[p]
            states.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()})
        out[tag] = dict(kwargs=kw, shapes=shapes, init=init, grads=grads, after=after, states=states)
        print("adafactor", tag, "lr of p0 after 3 steps:", opt._get_lr(opt.param_groups[0], opt.state[params[0]]))
    save_fixture(out, "adafactor_steps")


def conditioner_case():
    """GeneralConditioner with the SDXL cond_stage layout (modules/encoders/embedding.py:59-149, metadata.py:14-36): two
    pass-through embedders standing in for the frozen text encoders' outputs (token features -> "crossattn", pooled ->
    "vector") and three ConcatTimestepEmbedderND(256) for original size / crop / target size (lists of tuples, as the
    dataset provides them)."""
    from neurosis.modules.encoders.embedding import AbstractEmbModel, GeneralConditioner
    from neurosis.modules.encoders.metadata import ConcatTimestepEmbedderND

    class Passthrough(AbstractEmbModel):
        def forward(self, x):
            return x

    g = torch.Generator().manual_seed(99)
    batch = {
        "image": torch.zeros(3, 3, 8, 8),
        "tokens_l": torch.randn(3, 77, 48, generator=g),
        "tokens_g": torch.randn(3, 77, 80, generator=g),
        "pooled_g": torch.randn(3, 80, generator=g),
        "original_size_as_tuple": [(1024, 1024), (832, 1216), (1536, 640)],
        "crop_coords_top_left": [(0, 0), (64, 32), (7, 511)],
        "target_size_as_tuple": [(1024, 1024), (832, 1216), (1344, 768)],
    }
    cond = GeneralConditioner([
        Passthrough(input_key="tokens_l"), Passthrough(input_key="tokens_g"), Passthrough(input_key="pooled_g"),
        ConcatTimestepEmbedderND(256, input_key="original_size_as_tuple"), ConcatTimestepEmbedderND(256, input_key="crop_coords_top_left"),
        ConcatTimestepEmbedderND(256, input_key="target_size_as_tuple"),
    ])
    out = cond(batch)
    zero = cond(batch, force_zero_embeddings=["pooled_g", "crop_coords_top_left"])
    save_fixture(dict(batch=batch, out=out, zero=zero), "conditioner_sdxl")
    print("conditioner:", {k: tuple(v.shape) for k, v in out.items()})


def decoder_case(nmodel):
    """VAE decoder (post_quant_conv + decode) on tiny latents."""
    dec = nmodel.Decoder(**VAE_TINY).eval()
    shapes = {k: list(v.shape) for k, v in dec.state_dict().items()}
    dec.load_state_dict(synth_state_dict(shapes))
    g = torch.Generator().manual_seed(123)
    z = torch.randn(2, 4, 8, 8, generator=g)
    with torch.no_grad():
        image = dec(z)
    save_fixture(dict(cfg=VAE_TINY, z=z, image=image), "vae_decoder_tiny")
    (HERE / "vae_decoder_tiny_keys.json").write_text(json.dumps(shapes, indent=0))
    print("decoder: image", tuple(image.shape), float(image.abs().mean()))


VAE_GRAD_KEYS = [
    "encoder.conv_in.weight", "encoder.down.0.block.0.conv1.weight", "encoder.down.1.block.0.nin_shortcut.weight", "encoder.down.1.downsample.conv.weight",
    "encoder.mid.attn_1.q.weight", "encoder.mid.attn_1.k.weight", "encoder.norm_out.weight", "encoder.conv_out.bias",
    "encoder.quant_conv.weight", "decoder.post_quant_conv.weight", "decoder.conv_in.weight", "decoder.mid.attn_1.v.bias", "decoder.mid.block_2.norm2.bias",
    "decoder.up.1.upsample.conv.weight", "decoder.up.1.block.0.nin_shortcut.weight", "decoder.up.0.block.2.conv2.weight", "decoder.norm_out.bias",
    "decoder.conv_out.weight",
]


def vae_train_case(nmodel):
    """One reconstruction training step's forward + backward of the autoencoder (AutoencodingEngine.forward, autoencoder.py:
    222-225: encode -> DiagonalGaussianRegularizer(sample=True) -> decode; loss = mse(x, xrec) as the engine's simple-loss
    branch, and a second case with a KL term as GeneralLPIPSWithDiscriminator's regularization_weights would add it)."""
    from neurosis.modules.regularizers import DiagonalGaussianRegularizer

    enc, dec = nmodel.Encoder(**VAE_TINY).train(), nmodel.Decoder(**VAE_TINY).train()
    shapes = {f"encoder.{k}": list(v.shape) for k, v in enc.state_dict().items()}
    shapes.update({f"decoder.{k}": list(v.shape) for k, v in dec.state_dict().items()})
    sd = synth_state_dict(shapes)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    reg = DiagonalGaussianRegularizer(sample=True)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    out = {"cfg": VAE_TINY, "x": x, "cases": {}}
    for tag, kl_weight in (("rec_only", 0.0), ("rec_kl", 1e-2)):
        for p in list(enc.parameters()) + list(dec.parameters()):
            p.grad = None
        torch.manual_seed(2718)
        noise = torch.randn(2, 4, 8, 8)
        torch.manual_seed(2718)                   # posterior.sample() draws torch.randn(mean.shape) from the global generator
        moments = enc(x)
        z, log = reg(moments)
        xrec = dec(z)
        loss = torch.nn.functional.mse_loss(x, xrec) + kl_weight * log["kl_loss"]
        loss.backward()
        named = {f"encoder.{k}": p for k, p in enc.named_parameters()}
        named.update({f"decoder.{k}": p for k, p in dec.named_parameters()})
        out["cases"][tag] = dict(kl_weight=kl_weight, noise=noise, moments=moments.detach(), z=z.detach(), xrec=xrec.detach(), loss=loss.detach(),
                                 kl_loss=log["kl_loss"].detach(), grads={k: named[k].grad.clone() for k in VAE_GRAD_KEYS},
                                 grad_norms={k: float(p.grad.norm()) for k, p in named.items()})
        assert torch.allclose(z.detach(), moments[:, :4].detach() + torch.exp(0.5 * moments[:, 4:].detach().clamp(-30, 20)) * noise, atol=1e-6)
        print(f"vae train {tag}: loss={float(loss):.5f} kl={float(log['kl_loss']):.3f}")
    save_fixture(out, "vae_train_tiny")
    (HERE / "vae_train_tiny_keys.json").write_text(json.dumps(shapes, indent=0))


DISC_TINY = dict(input_nc=3, ndf=8, n_layers=3)


def disc_state_dict(shapes: dict) -> dict:
    """synthetic discriminator weights: every float tensor from synth_tensor, BatchNorm running_var kept positive"""
    sd = synth_state_dict({k: v for k, v in shapes.items() if not k.endswith("num_batches_tracked")})
    for k in sd:
        if k.endswith("running_var"):
            sd[k] = sd[k].abs() + 0.5
        elif k.replace(".weight", ".running_var") in sd and k.endswith(".weight"):
            sd[k] = 1.0 + 0.1 * sd[k]          # BatchNorm scales near 1, as the reference's weights_init draws them (N(1, 0.02))
    return sd


def discriminator_case():
    """The PatchGAN discriminator in training mode (modules/losses/patchgan/model.py) with the two discriminator losses
    (modules/losses/functions.py:21-50) as GeneralLPIPSWithDiscriminator's optimizer_idx == 1 branch calls them
    (discriminator_loss.py:303-320: D(real) then D(fake), both detached), and the generator's adversarial term
    g_loss = -mean(D(fake)) (:268-270) with its gradient w.r.t. the fake image."""
    from neurosis.modules.losses.functions import get_discr_loss_fn
    from neurosis.modules.losses.patchgan.model import NLayerDiscriminator

    disc = NLayerDiscriminator(**DISC_TINY).train()
    shapes = {k: list(v.shape) for k, v in disc.state_dict().items()}
    g = torch.Generator().manual_seed(404)
    real = torch.rand(4, 3, 64, 64, generator=g) * 2 - 1
    fake = (real + 0.3 * torch.randn(4, 3, 64, 64, generator=g)).clamp(-1, 1)
    out = {"cfg": DISC_TINY, "real": real, "fake": fake, "cases": {}}
    for kind in ("hinge", "vanilla"):
        disc.load_state_dict(disc_state_dict(shapes), strict=False)
        for b in disc.buffers():
            if b.dtype == torch.int64:
                b.zero_()
        for p in disc.parameters():
            p.grad = None
        logits_real = disc(real)
        logits_fake = disc(fake)
        d_loss = get_discr_loss_fn(kind)(logits_real, logits_fake)
        d_loss.backward()
        grads = {k: p.grad.clone() for k, p in disc.named_parameters()}
        buffers = {k: v.clone() for k, v in disc.state_dict().items() if "running" in k or "num_batches" in k}
        out["cases"][kind] = dict(logits_real=logits_real.detach(), logits_fake=logits_fake.detach(), d_loss=d_loss.detach(), grads=grads, buffers=buffers)
        print(f"discriminator {kind}: d_loss={float(d_loss.detach()):.5f} logits {tuple(logits_real.shape)}")
    # generator side: gradient of -mean(D(fake)) w.r.t. the image (fresh buffers, as a generator step would see them)
    disc.load_state_dict(disc_state_dict(shapes), strict=False)
    for p in disc.parameters():
        p.grad = None
    img = fake.clone().requires_grad_(True)
    g_loss = -disc(img).mean()
    g_loss.backward()
    out["generator"] = dict(g_loss=g_loss.detach(), d_image=img.grad.clone())
    save_fixture(out, "patchgan_tiny")
    (HERE / "patchgan_tiny_keys.json").write_text(json.dumps(shapes, indent=0))


def gan_generator_case(nmodel):
    """The generator (autoencoder) branch of the reference's GeneralLPIPSWithDiscriminator.forward (discriminator_loss.py:229-303), run by
    the reference itself on the tiny autoencoder + PatchGAN of the cases above.  As the reference's engine calls it -- weights=None -- the
    branch raises at :300 (`if weights > 0` with None) and the value it returns at :281 is an UN-REDUCED tensor (p_rec_loss is [B,C,H,W]), on
    which manual_backward cannot be called; with a tensor `weights = 1` the forward does run, and its pieces are what the fixture pins:
    nll_loss (:263), g_loss (:270) and the adaptive weight (:272, calculate_adaptive_weight :196-208) on the decoder's last layer.  The
    object is built without __init__ (which downloads LPIPS weights); perceptual_weight = 0 keeps LPIPS out of the branch."""
    import neurosis.modules.autoencoding.losses.discriminator_loss as ndl
    from neurosis.modules.losses.patchgan.model import NLayerDiscriminator
    from neurosis.modules.regularizers import DiagonalGaussianRegularizer

    hp = dict(disc_factor=0.7, disc_weight=0.9, rec_weight=0.02, logvar_init=2.5)       # (small nll gradients: the adaptive weight stays under its 1e4 clamp)
    enc, dec = nmodel.Encoder(**VAE_TINY).train(), nmodel.Decoder(**VAE_TINY).train()
    shapes = {f"encoder.{k}": list(v.shape) for k, v in enc.state_dict().items()}
    shapes.update({f"decoder.{k}": list(v.shape) for k, v in dec.state_dict().items()})
    sd = synth_state_dict(shapes)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")})
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    disc = NLayerDiscriminator(**DISC_TINY).train()
    dshapes = {k: list(v.shape) for k, v in disc.state_dict().items()}
    disc.load_state_dict(disc_state_dict(dshapes), strict=False)
    loss = ndl.GeneralLPIPSWithDiscriminator.__new__(ndl.GeneralLPIPSWithDiscriminator)
    torch.nn.Module.__init__(loss)
    loss.dims, loss.scale_input_to_tgt_size, loss.perceptual_weight, loss.learn_logvar = 2, False, 0.0, False
    loss.logvar = torch.nn.Parameter(torch.ones(size=()) * hp["logvar_init"])
    loss.discriminator, loss.disc_start, loss.disc_factor, loss.discriminator_weight = disc, 0, hp["disc_factor"], hp["disc_weight"]
    loss.rec_weight, loss.rec_loss_type, loss.regularization_weights, loss.additional_log_keys = hp["rec_weight"], "l2", {}, set()
    loss.train()
    g = torch.Generator().manual_seed(77)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1                 # (the image of vae_train_case)
    torch.manual_seed(2718)
    noise = torch.randn(2, 4, 8, 8)
    torch.manual_seed(2718)
    z, _ = DiagonalGaussianRegularizer(sample=True)(enc(x))
    xrec = dec(z)
    total, log = loss(x, xrec, global_step=1, optimizer_idx=0, last_layer=dec.get_last_layer(), weights=torch.tensor(1.0))
    out = dict(hp=hp, x=x, noise=noise, xrec=xrec.detach(), nll=log["train/loss/nll"], g_loss=log["train/loss/g"], d_weight=log["train/scalars/d_weight"],
               total_shape=list(total.shape), total_mean=log["train/loss/total"])
    save_fixture(out, "gan_generator_tiny")
    print(f"gan generator branch: nll={float(out['nll']):.4f} g={float(out['g_loss']):.5f} d_weight={float(out['d_weight']):.4f} total is {tuple(total.shape)}")


GLUE_SIGMAS = [0.002, 0.0292, 0.35, 0.9, 1.0, 2.7, 14.6, 80.0]
GLUE_T = [0.0, 0.0004, 0.1, 0.37, 0.5, 0.93, 0.9995]
GLUE_CLASSES = {
    "preconditioning": {"EpsPreconditioning": {}, "VPreconditioning": {}, "VPreconditioningWithEDMcNoise": {}, "EDMPreconditioning": {},
                        "EDMPreconditioning/0.5": dict(sigma_data=0.5), "RectifiedFlowXLPreconditioning": {}, "RectifiedFlowComfyPreconditioning": {}},
    "weighting": {"UnitWeighting": {}, "EpsWeighting": {}, "EDMWeighting": {}, "EDMWeighting/0.5": dict(sigma_data=0.5), "RectifiedFlowWeighting": {},
                  "RectifiedFlowWeighting/ms": dict(m=0.3, s=1.7), "RectifiedFlowComfyWeighting": {}, "RectifiedFlowComfyWeighting/ms": dict(m=-0.2, s=0.8)},
    "generator": {"EDMSigmaGenerator": {}, "CosineScheduleSigmaGenerator": {}, "TanScheduleSigmaGenerator": {}, "TanScheduleSigmaGenerator/noclip": dict(clip=False, scale=2.0),
                  "RectifiedFlowSigmaGenerator": {}, "RectifiedFlowComfySigmaGenerator": {}, "RectifiedFlowComfySigmaGenerator/shift": dict(start_shift=0.01, end_shift=0.02)},
    "discretization": {"EDMcDiscretization": {}, "EDMcSimpleDiscretization": {}, "RectifiedFlowDiscretization": {}, "RectifiedFlowComfyDiscretization": {},
                       "RectifiedFlowDiscretization/zero": dict(do_append_zero=True), "TanZeroSNRDiscretization": {}, "EDMDiscretization": {},
                       "LegacyDDPMDiscretization": {}},
}


def glue_class_cases(nd):
    """Every preconditioning / weighting / sigma-generator / discretisation class of rows A4-A5 on fixed inputs, plus the
    rectified-flow objective of StandardDiffusionLoss (loss.py:128-137) on the tiny SDXL-style UNet with injected sigma / noise."""
    import neurosis.modules.diffusion.denoiser_preconditioning as npre
    import neurosis.modules.diffusion.denoiser_weighting as nw
    import neurosis.modules.diffusion.discretization as ndisc
    import neurosis.modules.diffusion.sampling.sigma_generators as ng

    sig = torch.tensor(GLUE_SIGMAS)
    t = torch.tensor(GLUE_T, dtype=torch.float64)
    out = {k: {} for k in GLUE_CLASSES}
    for name, kw in GLUE_CLASSES["preconditioning"].items():
        out["preconditioning"][name] = [v.clone() for v in getattr(npre, name.split("/")[0])(**kw)(sig)]
    comfy_sig = torch.tensor([0.001, 0.1, 0.37, 0.5, 0.93, 0.999])
    for name, kw in GLUE_CLASSES["weighting"].items():
        out["weighting"][name] = getattr(nw, name.split("/")[0])(**kw)(comfy_sig if "Comfy" in name else sig)
    out["weighting"]["MinSNRGamma/eps"] = nw.MinSNRGammaModifier(nw.EpsWeighting(), gamma=5)(sig)
    out["weighting"]["MinSNRGamma/v"] = nw.MinSNRGammaModifier(nw.EDMWeighting(0.5), gamma=3, v_pred=True)(sig)
    for name, kw in GLUE_CLASSES["generator"].items():
        gen = getattr(ng, name.split("/")[0])(**kw)
        out["generator"][name] = gen(len(t), t.float() if "Cosine" in name else t)
    out["generator"]["CosineScheduleSigmaGenerator/shift"] = ng.CosineScheduleSigmaGenerator()(len(t), t.float(), shift=2, return_logSNR=True)
    spaced = ndisc.generate_roughly_equally_spaced_steps
    ndisc.generate_roughly_equally_spaced_steps = lambda n, m: np.ascontiguousarray(spaced(n, m))
    for name, kw in GLUE_CLASSES["discretization"].items():
        disc = getattr(ndisc, name.split("/")[0])(**kw)
        out["discretization"][name] = [disc(n).detach().clone() for n in (1000, 10)] + [disc(10, flip=True).detach().clone()]

    # rectified-flow objective
    torch.manual_seed(0)
    net = nd.UNetModel(**UNET_TINY).eval()
    shapes = json.loads((HERE / "unet_sdxl_tiny_keys.json").read_text())
    net.load_state_dict(synth_state_dict(shapes))
    fx = load_fixture("unet_sdxl_tiny")
    sigma = torch.tensor([0.35, 0.8])
    denoiser = nd.Denoiser(preconditioning=npre.RectifiedFlowComfyPreconditioning())
    weighting = nw.RectifiedFlowComfyWeighting()
    cond = {"crossattn": fx["context"], "vector": fx["y"]}
    s_bc = sigma[:, None, None, None]
    z_t = (1.0 - s_bc) * fx["x"] + s_bc * fx["noise"]
    eps_out = denoiser(nd.OpenAIWrapper(net), z_t, sigma, cond, "F")
    loss = ((eps_out.float() - fx["noise"].float()) ** 2).flatten(1).mean(1) * weighting(sigma).float()
    loss.mean().backward()
    out["rf"] = dict(sigma=sigma, z_t=z_t.detach(), F_out=eps_out.detach(), loss=loss.detach(),
                     grads={k: p.grad.detach().clone() for k, p in net.named_parameters() if k in GRAD_KEYS[:6]},
                     grad_norms={k: float(p.grad.norm()) for k, p in net.named_parameters()})
    save_fixture(out, "glue_classes")
    print("glue classes:", {k: len(v) for k, v in out.items()}, "rf loss", loss.tolist())


VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)


class _VGG16Taps(torch.nn.Module):
    """torchvision is not installed here: vgg16().features[:30] as plain torch modules under torchvision's parameter names, returning
    the five activations the reference taps through create_feature_extractor (perceptual.py:45-60): features.3/8/15/22/29."""

    def __init__(self):
        super().__init__()
        nn, layers, cin = torch.nn, [], 3
        for item in VGG16_CFG:
            if item == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, item, 3, padding=1), nn.ReLU(inplace=False)]
                cin = item
        self.features = nn.Sequential(*layers)

    def forward(self, x):
        out, names = {}, {3: "relu1", 8: "relu2", 15: "relu3", 22: "relu4", 29: "relu5"}
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in names:
                out[names[i]] = x
        return out


class _AlexTaps(torch.nn.Module):
    """alexnet().features[:12] as plain torch modules under torchvision's parameter names, returning the five activations the
    reference taps (perceptual.py:36-44): features.1/4/7/9/11."""

    def __init__(self):
        super().__init__()
        nn = torch.nn
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, 11, stride=4, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=False), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=False), nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=False),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=False))

    def forward(self, x):
        out, names = {}, {1: "relu1", 4: "relu2", 7: "relu3", 9: "relu4", 11: "relu5"}
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in names:
                out[names[i]] = x
        return out


def lpips_case():
    """The reference's LPIPS.forward (scaling layer, normalize_tensor, calibrated lin layers from its package data, spatial average)
    over the VGG16 / AlexNet stand-ins above with synthetic trunk weights, and its gradient w.r.t. the second image."""
    import neurosis.modules.losses.perceptual as nper

    for kind, trunk, gain, hw, seed in (("vgg", _VGG16Taps().eval(), 1.6, (64, 64), 909), ("alex", _AlexTaps().eval(), 1.6, (96, 80), 910)):
        shapes = {f"pnet.{k}": list(v.shape) for k, v in trunk.state_dict().items()}
        trunk.load_state_dict({k[len("pnet."):]: v * gain for k, v in synth_state_dict(shapes).items()})   # (gain: keeps ReLU activations from dying out)
        lp = nper.LPIPS.__new__(nper.LPIPS)
        torch.nn.Module.__init__(lp)
        lp.pnet_type, lp.lpips, lp.spatial, lp.pnet = kind, True, False, trunk
        lp.pnet_conf = nper.PNET_CONFIG[kind]
        lp.chns, lp.out_type = lp.pnet_conf["channels"], lp.pnet_conf["out_type"]
        lp.L, lp.pnet_keys = len(lp.chns), list(lp.pnet_conf["features"].values())
        lp.scaling_layer = nper.ScalingLayer()
        lins = [nper.NetLinLayer(c) for c in lp.chns]
        lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4 = lins
        lp.lins = torch.nn.ModuleDict(dict(zip(lp.pnet_keys, lins)))
        lp._load_pretrained(kind)
        lp.requires_grad_(False)
        g = torch.Generator().manual_seed(seed)
        x = torch.rand(2, 3, *hw, generator=g) * 2 - 1
        y = (x + 0.25 * torch.randn(2, 3, *hw, generator=g)).clamp(-1, 1).requires_grad_(True)
        dist = lp(x, y)
        (dist.reshape(-1) * torch.tensor([1.0, 0.5])).sum().backward()
        lin = {k: v.clone() for k, v in lp.state_dict().items() if k.startswith("lin") and not k.startswith("lins")}
        save_fixture(dict(x=x, y=y.detach(), distance=dist.detach(), upstream=torch.tensor([1.0, 0.5]), d_y=y.grad.clone(), lin=lin), f"lpips_{kind}_tiny")
        (HERE / f"lpips_{kind}_tiny_keys.json").write_text(json.dumps(shapes, indent=0))
        print(f"lpips[{kind}]:", dist.reshape(-1).tolist(), "grad norm", float(y.grad.norm()))


def analytic_denoiser(x, sigma, c, *args, **kwargs):
    """A closed-form stand-in for denoiser(network, ...) so that sampler arithmetic can be pinned without a network:
    depends on x, on sigma and (through "vector") on the conditioning, so guidance has something to act on."""
    shrink = 1.0 / (1.0 + sigma**2)
    return x * shrink[:, None, None, None] + 0.1 * torch.tanh(c["vector"])[:, :, None, None] * (sigma / (1.0 + sigma))[:, None, None, None]


SAMPLER_CASES = {
    # name: (class name, init kwargs, guidance scale or None, steps)
    "euler": ("EulerEDMSampler", {}, None, 7),
    "euler_cfg": ("EulerEDMSampler", {}, 5.0, 7),
    "euler_churn_cfg": ("EulerEDMSampler", {"s_churn": 2.0, "s_tmin": 0.5, "s_tmax": 10.0, "s_noise": 1.003}, 3.0, 6),
    "heun_cfg": ("HeunEDMSampler", {}, 4.0, 5),
    "euler_a_cfg": ("EulerAncestralSampler", {"eta": 0.8, "s_noise": 0.9}, 6.0, 7),
    "dpmpp2s_a": ("DPMPP2SAncestralSampler", {}, None, 6),
    "dpmpp2m_cfg": ("DPMPP2MSampler", {}, 7.5, 8),
    "lms_cfg": ("LinearMultistepSampler", {"order": 3}, 2.0, 6),
}


def sampler_inputs():
    g = torch.Generator().manual_seed(2024)
    x0 = torch.randn(3, 4, 6, 5, generator=g)
    cond = {"vector": torch.randn(3, 4, generator=g)}
    uc = {"vector": torch.randn(3, 4, generator=g)}
    return x0, cond, uc


def seeded_noise_sampler(seed: int):
    g = torch.Generator().manual_seed(seed)
    return lambda x: torch.randn(x.shape, generator=g).to(x)


def sampler_cases(nd):
    """Every sampler class against the analytic denoiser (pins the sampler arithmetic), then Euler+CFG and Heun+CFG
    trajectories through the tiny SDXL UNet with the reference's DiscreteDenoiser (pins the whole sampling path)."""
    import neurosis.modules.diffusion.discretization as ndisc
    import neurosis.modules.diffusion.sampling as ns
    import numpy as np
    from neurosis.modules.guidance import VanillaCFG

    # torch >= 2.x refuses to index with the negative-stride numpy view the reference builds for n < 1000
    # ("At least one stride in the given numpy array is negative"): hand it the same indices as a contiguous array
    spaced = ndisc.generate_roughly_equally_spaced_steps
    ndisc.generate_roughly_equally_spaced_steps = lambda n, m: np.ascontiguousarray(spaced(n, m))

    out = {}
    for name, (cls, kwargs, scale, steps) in SAMPLER_CASES.items():
        x0, cond, uc = sampler_inputs()
        sampler = getattr(ns, cls)(discretization=nd.LegacyDDPMDiscretization(), guider=None if scale is None else VanillaCFG(scale), num_steps=steps,
                                   device="cpu", **kwargs)
        if hasattr(sampler, "noise_sampler"):
            sampler.noise_sampler = seeded_noise_sampler(77)
        torch.manual_seed(4321)                      # the churn noise comes from the global generator (torch.randn_like)
        with torch.no_grad():
            out[name] = sampler(analytic_denoiser, x0.clone(), cond, uc=uc)
        print(f"sampler {name}: |x|={float(out[name].abs().mean()):.5f}")
    save_fixture(out, "sampler_analytic")

    torch.manual_seed(0)
    net = nd.UNetModel(**UNET_TINY).eval()
    shapes = json.loads((HERE / "unet_sdxl_tiny_keys.json").read_text())
    net.load_state_dict(synth_state_dict(shapes))
    denoiser = nd.DiscreteDenoiser(preconditioning=nd.EpsPreconditioning(), num_idx=1000, discretization=nd.LegacyDDPMDiscretization())
    wrapper = nd.OpenAIWrapper(net)
    g = torch.Generator().manual_seed(555)
    B = 2
    noise = torch.randn(B, 4, 16, 16, generator=g)
    cond = {"crossattn": torch.randn(B, 7, UNET_TINY["context_dim"], generator=g), "vector": torch.randn(B, UNET_TINY["adm_in_channels"], generator=g)}
    uc = {"crossattn": torch.randn(B, 7, UNET_TINY["context_dim"], generator=g), "vector": torch.zeros(B, UNET_TINY["adm_in_channels"])}

    def denoiser_cb(inputs, sigma, c):
        return denoiser(wrapper, inputs, sigma, c, "D")

    fixture = dict(noise=noise, cond=cond, uc=uc, runs={})
    for name, cls, steps, scale in (("euler_cfg", "EulerEDMSampler", 4, 5.0), ("heun_cfg", "HeunEDMSampler", 3, 3.0), ("euler_plain", "EulerEDMSampler", 3, None)):
        sampler = getattr(ns, cls)(discretization=nd.LegacyDDPMDiscretization(), guider=None if scale is None else VanillaCFG(scale), num_steps=steps, device="cpu")
        trajectory = []
        step = sampler.sampler_step

        def recording_step(*a, _step=step, **k):
            x = _step(*a, **k)
            trajectory.append(x.detach().clone())
            return x

        sampler.sampler_step = recording_step
        with torch.no_grad():
            final = sampler(denoiser_cb, noise.clone(), cond, uc=uc)
        fixture["runs"][name] = dict(cls=cls, steps=steps, scale=scale, trajectory=trajectory, final=final)
        print(f"unet sampler {name}: |x|={float(final.abs().mean()):.5f} steps={len(trajectory)}")
    save_fixture(fixture, "sampler_unet_tiny")


HF_CLIP_TINY = dict(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4, max_position_embeddings=77,
                    hidden_act="quick_gelu", eos_token_id=2, bos_token_id=0, pad_token_id=1)
OPENCLIP_TINY = dict(vocab_size=1000, width=128, layers=3, heads=2, context_length=77, embed_dim=96)


def clip_token_ids():
    """[3, 77] ids in the CLIP convention: BOS = vocab-2, EOS = vocab-1 (the highest id), padding after EOS = EOS (bigG) --
    prompts of different lengths, one filling the whole context."""
    g = torch.Generator().manual_seed(31)
    ids = torch.randint(3, 990, (3, 77), generator=g)
    ids[:, 0] = 998
    for row, end in enumerate((9, 40, 76)):
        ids[row, end:] = 999
    return ids


class _OpenCLIPTextStandIn(torch.nn.Module):
    """open_clip is not installed here (reference pyproject: open-clip-torch >= 2.2.0).  This is its published text tower --
    open_clip/transformer.py ResidualAttentionBlock (ln_1 -> nn.MultiheadAttention with the additive causal mask -> residual,
    ln_2 -> c_fc, GELU, c_proj -> residual) under open_clip.CLIP's attribute names -- so that the REFERENCE's own
    encode_with_transformer / text_transformer_forward / pool (models/text_encoder/clip.py:311-343) can run over it."""

    class Block(torch.nn.Module):
        def __init__(self, width, heads):
            super().__init__()
            nn = torch.nn
            self.ln_1, self.ln_2 = nn.LayerNorm(width), nn.LayerNorm(width)
            self.attn = nn.MultiheadAttention(width, heads)
            self.mlp = nn.Sequential()
            self.mlp.add_module("c_fc", nn.Linear(width, 4 * width))
            self.mlp.add_module("gelu", nn.GELU())
            self.mlp.add_module("c_proj", nn.Linear(4 * width, width))

        def forward(self, x, attn_mask=None):
            h = self.ln_1(x)
            x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
            return x + self.mlp(self.ln_2(x))

    def __init__(self, vocab_size, width, layers, heads, context_length, embed_dim):
        super().__init__()
        nn = torch.nn
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(self.Block(width, heads) for _ in range(layers))
        self.transformer.grad_checkpointing = False
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]))
        mask = torch.empty(context_length, context_length).fill_(float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)


def _bare(cls):
    """an instance of a reference embedder without running its constructor (which downloads tokenizers / weights)"""
    obj = cls.__new__(cls)
    torch.nn.Module.__init__(obj)
    obj.ucg_rate, obj.is_trainable, obj.extended_chunks, obj.max_length, obj.device = 0.0, False, 0, 77, "cpu"
    return obj


def text_encoder_cases():
    """The reference's FrozenCLIPEmbedder.forward over transformers' CLIPTextModel (tiny, seeded weights) and its
    FrozenOpenCLIPEmbedder2.forward over the open_clip stand-in above, on fixed token ids (the tokenizer is replaced by a
    function returning them: the CLIP vocabulary is not on this machine)."""
    import neurosis.models.text_encoder.clip as nclip
    from transformers import CLIPTextModel
    from transformers.models.clip import CLIPTextConfig

    ids = clip_token_ids()
    tokenizer = lambda text, **kw: {"input_ids": ids[: len(text)]}  # noqa: E731
    prompts = ["a", "b", "c"]

    hf = CLIPTextModel(CLIPTextConfig(**HF_CLIP_TINY)).eval()
    hf_shapes = {k: list(v.shape) for k, v in hf.state_dict().items()}
    hf.load_state_dict(synth_state_dict(hf_shapes))
    out = {"ids": ids, "hf_cfg": HF_CLIP_TINY, "openclip_cfg": OPENCLIP_TINY, "hf": {}, "openclip": {}}
    for tag, layer, layer_idx, pooled in (("hidden11_style", "hidden", 3, False), ("penultimate_pooled", "penultimate", None, True),
                                          ("hidden_neg", "hidden", 1, True)):
        emb = _bare(nclip.FrozenCLIPEmbedder)
        emb.transformer, emb.tokenizer = hf, tokenizer
        emb.layer, emb.return_pooled, emb.output_hidden_states = layer, pooled, True
        emb.layer_idx = 10 if layer == "penultimate" else layer_idx        # what the reference constructor assigns
        if layer == "penultimate":
            emb.layer_idx = 2                                              # ... scaled to this 4-layer model: depth - 2
        with torch.no_grad():
            res = emb(prompts)
        out["hf"][tag] = dict(layer=layer, layer_idx=emb.layer_idx, return_pooled=pooled, result=res)
    with torch.no_grad():
        full = hf(input_ids=ids, output_hidden_states=True)
    out["hf"]["raw"] = dict(last_hidden_state=full.last_hidden_state, pooler_output=full.pooler_output, hidden_states=list(full.hidden_states))

    oc = _OpenCLIPTextStandIn(**OPENCLIP_TINY).eval()
    oc_shapes = {k: list(v.shape) for k, v in oc.state_dict().items()}
    oc.load_state_dict(synth_state_dict(oc_shapes))
    for tag, layer, pooled, legacy in (("penultimate_pooled", "penultimate", True, False), ("last", "last", False, False), ("legacy_last", "last", False, True),
                                       ("pooled_layer", "pooled", False, False)):
        emb = _bare(nclip.FrozenOpenCLIPEmbedder2)
        emb.model, emb.tokenizer = oc, tokenizer
        emb.layer, emb.return_pooled, emb.legacy = layer, pooled, legacy
        with torch.no_grad():
            res = emb(prompts)
        out["openclip"][tag] = dict(layer=layer, return_pooled=pooled, legacy=legacy, result=res)
    save_fixture(out, "text_encoders_tiny")
    (HERE / "text_encoders_tiny_keys.json").write_text(json.dumps({"hf": hf_shapes, "openclip": oc_shapes}, indent=0))
    print("text encoders: hf", tuple(full.last_hidden_state.shape), "openclip pooled", tuple(out["openclip"]["penultimate_pooled"]["result"][1].shape))


BUCKET_LIST_CASES = {
    "default": {}, "n15": dict(n_buckets=15), "n9_atan": dict(n_buckets=9, use_atan=True), "interp": dict(n_buckets=15, bias_square=False),
    "small": dict(n_buckets=7, edge_min=256, edge_max=1024, edge_step=64, max_aspect=2.0, tgt_pixels=512 * 512, tolerance=10),
}
BUCKET_RATIOS = [0.2, 0.25, 0.3333, 0.5, 0.5625, 0.6667, 0.75, 0.8, 0.95, 1.0, 1.05, 1.25, 1.3333, 1.5, 1.7778, 2.0, 2.3, 3.0, 4.0]


def bucket_assignment(n=400):
    """a synthetic dataset: the bucket index of each of n samples (some buckets smaller than a batch)"""
    g = np.random.default_rng(5)
    return g.choice([3, 7, 8, 12, 20, 21, 33], size=n, p=[0.01, 0.2, 0.15, 0.3, 0.25, 0.08, 0.01]).astype(np.int32)


def dataset_cases():
    """Aspect buckets, the bucketed batch schedule and the distributed sampler (SURVEY N4 data side)."""
    import pandas as pd
    from neurosis.dataset.aspect.bucket import AspectBucketList
    from neurosis.dataset.aspect.lists import SDXLBucketList
    from neurosis.dataset.aspect.sampler import AspectDistributedSampler
    from neurosis.dataset.imagefolder.aspect import ImageFolderDataset

    out = {"lists": {}, "lookup": {}, "schedule": {}, "sampler": {}}
    for name, kw in BUCKET_LIST_CASES.items():
        try:
            lst = AspectBucketList(**kw)
        except ValueError as err:          # (the reference's own defaults ask for more buckets than its constraints yield)
            out["lists"][name] = out["lookup"][name] = f"ValueError: {err}"
            continue
        out["lists"][name] = [(b.width, b.height, b.error) for b in lst]
        out["lookup"][name] = [int(lst.bucket_idx(r)) for r in BUCKET_RATIOS]
    for name, kw in (("sdxl", {}), ("sdxl_atan", dict(use_atan=True)), ("sdxl_interp", dict(bias_square=False))):
        lst = SDXLBucketList(**kw)
        out["lists"][name] = [(b.width, b.height, b.error) for b in lst]
        out["lookup"][name] = [int(lst.bucket_idx(r)) for r in BUCKET_RATIOS]

    class _Holder:       # what get_batch_iterator reads from `self`
        pass

    for batch_size in (4, 16):
        holder = _Holder()
        holder.samples, holder.batch_size = pd.DataFrame({"bucket_idx": bucket_assignment()}), batch_size
        np.random.seed(1234)
        out["schedule"][batch_size] = list(ImageFolderDataset.get_batch_iterator(holder))

    class _Batches:
        def get_batch_iterator(self):
            return iter(out["schedule"][4])

    for world, drop_last, shuffle in ((2, False, True), (8, False, True), (8, True, True), (3, False, False)):
        per_rank = []
        for rank in range(world):
            sampler = AspectDistributedSampler(_Batches(), num_replicas=world, rank=rank, shuffle=shuffle, seed=11, drop_last=drop_last)
            epochs = []
            for epoch in (0, 3):
                sampler.set_epoch(epoch)
                epochs.append(list(sampler))
            per_rank.append(epochs)
        out["sampler"][(world, drop_last, shuffle)] = per_rank
    save_fixture(out, "dataset_aspect")
    print("dataset: lists", {k: len(v) for k, v in out["lists"].items()}, "batches", {k: len(v) for k, v in out["schedule"].items()})


# ------------------------------------------------------------------------------------------------
# round 2: the loss CLASS itself, the engine's own methods, and the example configs' class-path tree
# ------------------------------------------------------------------------------------------------
class FixedSigma:
    """sigma generator stand-in handed to the reference's StandardDiffusionLoss: returns the injected sigmas whatever `t` is
    (the config's own DiscreteSigmaGenerator always yields 0 under the loss's t ~ U[0,1))."""

    def __init__(self, sigmas):
        self.sigmas = sigmas

    def __call__(self, n_samples, t=None):
        return self.sigmas[:n_samples].clone()


LOSS_CLASS_CASES = [("edm_l2", dict(loss_type="l2", objective_type="edm")), ("edm_l1", dict(loss_type="l1", objective_type="edm")),
                    ("edm_l2_offset", dict(loss_type="l2", objective_type="edm", noise_offset=0.1, noise_offset_chance=1.0)),
                    ("rf_l2", dict(loss_type="l2", objective_type="rf"))]


def loss_class_case(nd):
    """The reference's own `StandardDiffusionLoss._forward` / `get_loss` / `apply_noise_offset` (modules/diffusion/loss.py:
    32-40,105-157) executed on the tiny SDXL-shaped UNet -- round 1's fixture restated the loss assembly by hand.  The
    random draws of `_forward` (t, the noise, the per-(sample, channel) offset) come from the global generator after
    `torch.manual_seed(SEED)`; the fixture stores the tensors they produced so the HIP path can be fed the same ones."""
    from neurosis.modules.diffusion.loss import StandardDiffusionLoss

    SEED = 4242
    cfg = UNET_TINY
    torch.manual_seed(0)
    net = nd.UNetModel(**cfg).eval()
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth_state_dict(shapes))
    g = torch.Generator().manual_seed(77)
    B, HW = 2, 16
    x = torch.randn(B, 4, HW, HW, generator=g)
    ctx = torch.randn(B, 7, cfg["context_dim"], generator=g)
    y = torch.randn(B, cfg["adm_in_channels"], generator=g)
    out = {"cfg": cfg, "x": x, "context": ctx, "y": y, "seed": SEED, "cases": {}}
    for tag, kw in LOSS_CLASS_CASES:
        sig = torch.tensor([0.6, 0.3]) if kw["objective_type"] == "rf" else torch.tensor([0.8, 4.2])
        if kw["objective_type"] == "rf":
            denoiser = nd.Denoiser(preconditioning=nd.RectifiedFlowXLPreconditioning())
            weighting = nd.RectifiedFlowWeighting() if hasattr(nd, "RectifiedFlowWeighting") else nd.UnitWeighting()
        else:
            denoiser = nd.DiscreteDenoiser(preconditioning=nd.EpsPreconditioning(), num_idx=1000, discretization=nd.LegacyDDPMDiscretization())
            weighting = nd.EpsWeighting()
        loss_fn = StandardDiffusionLoss(sigma_generator=FixedSigma(sig), loss_weighting=weighting, **kw)
        for p_ in net.parameters():
            p_.grad = None
        # replay of the draws _forward makes, in its order, to record them
        torch.manual_seed(SEED)
        _t = torch.rand((B,), dtype=torch.float64)
        noise = torch.randn_like(x)
        offset = torch.randn(x.shape[:2] + (1, 1)) if kw.get("noise_offset", 0.0) > 0 else None
        torch.manual_seed(SEED)
        loss, extra = loss_fn._forward(nd.OpenAIWrapper(net), denoiser, {"crossattn": ctx, "vector": y}, x, {}, return_dict=True)
        loss.mean().backward()
        assert torch.equal(extra["t"], _t)
        grads = {k: p_.grad.detach().clone() for k, p_ in net.named_parameters() if k in GRAD_KEYS}
        gnorm = {k: float(p_.grad.norm()) for k, p_ in net.named_parameters()}
        # apply_noise_offset on its own, seeded, for the CPU check of this package's method
        torch.manual_seed(SEED + 1)
        offset_out = loss_fn.apply_noise_offset(noise.clone(), x)
        out["cases"][tag] = dict(kwargs=kw, sigma=sig, noise=noise, offset=offset, loss=loss.detach(), sigmas_out=extra["sigmas"].detach(), grads=grads,
                                 grad_norms=gnorm, offset_seed=SEED + 1, offset_out=offset_out, weighting=type(weighting).__name__,
                                 denoiser="rf" if kw["objective_type"] == "rf" else "discrete_eps")
        print(f"loss_class {tag}: loss={loss.tolist()}")
    save_fixture(out, "loss_class_tiny")


def engine_case(nd, nmodel):
    """The reference's own `DiffusionEngine` (models/diffusion.py:35-233) with the Lightning base class stubbed: constructor
    wiring (`_init_first_stage` with ddconfig.standalone=true, quirk Q4), `get_input`, `encode_first_stage` (chunked by
    vae_batch_size) and `training_step` on a tiny SDXL-shaped UNet + tiny VAE.  The draws of the loss come from the global
    generator after manual_seed(SEED), recorded as in loss_class_case."""
    import neurosis.models.autoencoder as nma
    import neurosis.models.diffusion as nmd
    from neurosis.modules.diffusion.loss import StandardDiffusionLoss

    SEED = 999
    torch.manual_seed(0)
    net = nd.UNetModel(**UNET_TINY).eval()
    ushapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth_state_dict(ushapes))
    vae = nma.AutoencoderKL(embed_dim=4, ddconfig={k: v for k, v in VAE_TINY.items() if k != "embed_dim"})
    vshapes = {k: list(v.shape) for k, v in vae.state_dict().items()}
    vae.load_state_dict(synth_state_dict(vshapes))
    denoiser = nd.DiscreteDenoiser(preconditioning=nd.EpsPreconditioning(), num_idx=1000, discretization=nd.LegacyDDPMDiscretization())
    sig = torch.tensor([1.7, 0.25, 6.0])

    class BatchCond(torch.nn.Module):          # conditioner stand-in: the batch already carries the conditioning tensors
        embedders = []

        def forward(self, batch, force_zero_embeddings=None):
            return {"crossattn": batch["crossattn"], "vector": batch["vector"]}

    loss_fn = StandardDiffusionLoss(sigma_generator=FixedSigma(sig), loss_weighting=nd.EpsWeighting())
    # the stubbed LightningModule has no trainer / logger plumbing: give the instance the three members training_step touches
    nmd.DiffusionEngine.loggers = property(lambda self: [])
    nmd.DiffusionEngine.save_hyperparameters = lambda self, *a, **k: None
    nmd.DiffusionEngine.global_step = 0
    logged = {}
    nmd.DiffusionEngine.log_dict = lambda self, d, **k: logged.update({kk: vv.detach().clone() for kk, vv in d.items()})
    eng = nmd.DiffusionEngine(model=net, denoiser=denoiser, first_stage_model=vae, conditioner=BatchCond(), sampler=None, optimizer=None, scheduler=None,
                              loss_fn=loss_fn, scale_factor=0.13025, input_key="image", vae_batch_size=2)
    g = torch.Generator().manual_seed(31)
    B = 3
    image = torch.rand(B, 3, 64, 64, generator=g) * 2 - 1
    batch = {"image": image, "crossattn": torch.randn(B, 7, UNET_TINY["context_dim"], generator=g), "vector": torch.randn(B, UNET_TINY["adm_in_channels"], generator=g)}
    with torch.no_grad():
        latents = eng.encode_first_stage(eng.get_input(batch))
    torch.manual_seed(SEED)
    _t = torch.rand((B,), dtype=torch.float64)
    noise = torch.randn_like(latents)
    torch.manual_seed(SEED)
    loss_mean = eng.training_step(dict(batch), 0)
    loss_mean.backward()
    grads = {k: p_.grad.detach().clone() for k, p_ in net.named_parameters() if k in GRAD_KEYS}
    sd_keys = sorted(k for k in eng.state_dict().keys())
    save_fixture(dict(unet_cfg=UNET_TINY, vae_cfg=VAE_TINY, image=image, crossattn=batch["crossattn"], vector=batch["vector"], sigma=sig, noise=noise,
                    latents=latents, loss_mean=loss_mean.detach(), logged=logged, grads=grads, scale_factor=0.13025, vae_batch_size=2, seed=SEED), "engine_tiny")
    (HERE / "engine_tiny_keys.json").write_text(json.dumps({"unet": ushapes, "vae": vshapes, "engine_state_dict_keys": sd_keys}, indent=0))
    print(f"engine: latents {tuple(latents.shape)} loss_mean={float(loss_mean):.6f} logged={ {k: float(v) for k, v in logged.items()} } state_dict keys={len(sd_keys)}")


# Single blocks at REAL SDXL / SD-VAE widths and small spatial size ("single-op vectors at real channel counts"): the tiny
# networks above run every HIP kernel at model_channels = 32 / head dim 16; these pin head dim 64, the 160-wide GEMM tiles, the halo
# convolution's 64-channel slabs, 640 / 1280-wide LayerNorm + GEGLU and the 512-channel VAE attention to the reference's own classes.
BLOCK_CASES = {
    # name: (kind, constructor kwargs, input shapes)
    "resblock_320_640": ("resblock", dict(channels=320, emb_channels=1280, dropout=0.0, out_channels=640, use_checkpoint=False), dict(x=(1, 320, 16, 16), emb=(1, 1280))),
    "tblock_640": ("tblock", dict(dim=640, n_heads=10, d_head=64, context_dim=2048, gated_ff=True, checkpoint=False, attn_mode="torch-sdp"),
                   dict(x=(1, 256, 640), context=(1, 77, 2048))),
    "tblock_1280": ("tblock", dict(dim=1280, n_heads=20, d_head=64, context_dim=2048, gated_ff=True, checkpoint=False, attn_mode="torch-sdp"),
                    dict(x=(1, 64, 1280), context=(1, 77, 2048))),
    "spatial_640_depth2": ("spatial", dict(in_channels=640, n_heads=10, d_head=64, depth=2, context_dim=2048, use_linear=True, attn_type="torch-sdp", use_checkpoint=False),
                           dict(x=(1, 640, 16, 16), context=(1, 77, 2048))),
    "vae_resnet_128_256": ("vae_resnet", dict(in_channels=128, out_channels=256, temb_channels=0, dropout=0.0), dict(x=(1, 128, 16, 16))),
    "vae_attn_512": ("vae_attn", dict(in_channels=512), dict(x=(1, 512, 16, 16))),
}
BLOCK_SAMPLE_ROWS = 8       # rows of every >= 2-D weight gradient that are stored (the rest is pinned through its norm)


def block_inputs(name: str, shapes: dict) -> dict:
    """bf16-exact inputs, a pure function of (case name, tensor name, shape): the tests rebuild them instead of storing them"""
    out = {}
    for k, shp in shapes.items():
        g = torch.Generator().manual_seed(zlib.crc32(f"{name}/{k}".encode()) & 0x7FFFFFFF)
        out[k] = torch.randn(*shp, generator=g).to(torch.bfloat16).to(torch.float32)
    return out


def block_upstream(name: str, shape) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(f"{name}/dy".encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g).to(torch.bfloat16).to(torch.float32)


def blocks_case(nmodel):
    """outputs, input gradients, every parameter's gradient norm and the first rows of every weight gradient of the reference's ResBlock
    (openaimodel.py:200-342), BasicTransformerBlock / SpatialTransformer (attention.py:420-511, 567-667), VAE ResnetBlock and AttnBlock
    (model.py:85-134, 144-173) at SDXL widths; written as .safetensors (no pickle)."""
    from safetensors.torch import save_file

    from neurosis.modules.attention import BasicTransformerBlock, SpatialTransformer
    from neurosis.modules.diffusion.openaimodel import ResBlock

    ctor = {"resblock": ResBlock, "tblock": BasicTransformerBlock, "spatial": SpatialTransformer, "vae_resnet": nmodel.ResnetBlock, "vae_attn": nmodel.AttnBlock}
    tensors, key_shapes = {}, {}
    for name, (kind, kw, in_shapes) in BLOCK_CASES.items():
        blk = ctor[kind](**kw).eval()
        shapes = {k: list(v.shape) for k, v in blk.state_dict().items()}
        blk.load_state_dict(synth_state_dict(shapes))
        ins = {k: v.clone().requires_grad_(True) for k, v in block_inputs(name, in_shapes).items()}
        if kind == "resblock":
            out = blk(ins["x"], ins["emb"])
        elif kind in ("tblock", "spatial"):
            out = blk(ins["x"], context=ins["context"])
        elif kind == "vae_resnet":
            out = blk(ins["x"], None)
        else:
            out = blk(ins["x"])
        dy = block_upstream(name, out.shape)
        out.backward(dy)
        tensors[f"{name}/out"] = out.detach().contiguous()
        for k, v in ins.items():
            tensors[f"{name}/d_{k}"] = v.grad.detach().contiguous()
        names = [k for k, _ in blk.named_parameters()]
        tensors[f"{name}/grad_norms"] = torch.tensor([float(p.grad.norm()) for _, p in blk.named_parameters()])
        for k, p in blk.named_parameters():
            g = p.grad.detach()
            tensors[f"{name}/g/{k}"] = (g[:BLOCK_SAMPLE_ROWS] if g.dim() >= 2 else g).contiguous()
        key_shapes[name] = dict(shapes=shapes, params=names)
        print(f"block {name}: out {tuple(out.shape)} |out| {float(out.abs().mean()):.4f} params {sum(p.numel() for p in blk.parameters())}")
    save_file(tensors, str(HERE / "blocks_real_width.safetensors"))
    (HERE / "blocks_real_width_keys.json").write_text(json.dumps(key_shapes, indent=0))
    print("blocks fixture bytes:", (HERE / "blocks_real_width.safetensors").stat().st_size)


def config_case():
    """The `model:` tree of the reference's example configs as DATA: for every node that names a class, where it sits, its
    class_path, the names of its init_args (and their values when they are plain scalars / lists of scalars), and whether that
    class path resolves in the reference itself.  tests/test_config_classpaths.py walks this with the `neurosis.` ->
    `neurosis_amd.` prefix swap INTEGRATION.md promises.  (Data derived from the YAML, not the YAML text.)"""
    import importlib

    import yaml

    def resolves(cp: str) -> bool:
        mod, _, name = cp.rpartition(".")
        try:
            return hasattr(importlib.import_module(mod), name)
        except Exception:
            return False

    def plain(v):
        if isinstance(v, dict):      # e.g. ddconfig: a dict of scalars / lists, no nested class
            return "class_path" not in v and all(plain(e) for e in v.values())
        return isinstance(v, (int, float, str, bool, type(None))) or (isinstance(v, list) and all(isinstance(e, (int, float, str, bool)) for e in v))

    out = {}
    for cfg_path in ("configs/sdxl/sdxl.example.yaml", "configs/sd15/sd15.example.yml"):
        cfg = yaml.safe_load(open(Path("/root/reference") / cfg_path))
        nodes = []

        def walk(node, where):
            if isinstance(node, dict):
                if "class_path" in node:
                    ia = node.get("init_args", {}) or {}
                    nodes.append({"where": where, "class_path": node["class_path"], "init_arg_names": sorted(ia.keys()),
                                  "plain_init_args": {k: v for k, v in ia.items() if plain(v)}, "resolves_in_reference": resolves(node["class_path"])})
                    for k, v in ia.items():
                        walk(v, f"{where}.init_args.{k}")
                else:
                    for k, v in node.items():
                        walk(v, f"{where}.{k}")
            elif isinstance(node, list):
                for i, v in enumerate(node):
                    walk(v, f"{where}[{i}]")

        walk(cfg["model"], "model")
        out[cfg_path] = {"nodes": nodes, "trainer": {k: cfg["trainer"].get(k) for k in ("precision", "accumulate_grad_batches", "strategy", "devices")}}
        print(f"config {cfg_path}: {len(nodes)} class nodes; unresolved in the reference itself: {[n['class_path'] for n in nodes if not n['resolves_in_reference']]}")
    (HERE / "config_class_paths.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = set(sys.argv[1:]) or {"unet", "vae", "glue", "adafactor", "conditioner", "decoder", "sampler", "text", "dataset", "vae_train", "disc", "glue_classes", "lpips", "loss_class", "engine", "config", "blocks", "gan_generator"}
    nd, nmodel = import_reference()
    if "unet" in which:
        unet_case(nd, UNET_TINY, "unet_sdxl_tiny", B=2, HW=16, with_y=True)
        unet_case(nd, UNET_SD15_TINY, "unet_sd15_tiny", B=2, HW=16, with_y=False)
    if "vae" in which:
        vae_case(nmodel)
    if "glue" in which:
        op_cases(nd)
    if "adafactor" in which:
        adafactor_case()
    if "conditioner" in which:
        conditioner_case()
    if "decoder" in which:
        decoder_case(nmodel)
    if "sampler" in which:
        sampler_cases(nd)
    if "text" in which:
        text_encoder_cases()
    if "dataset" in which:
        dataset_cases()
    if "vae_train" in which:
        vae_train_case(nmodel)
    if "disc" in which:
        discriminator_case()
    if "glue_classes" in which:
        glue_class_cases(nd)
    if "lpips" in which:
        lpips_case()
    if "loss_class" in which:
        loss_class_case(nd)
    if "engine" in which:
        engine_case(nd, nmodel)
    if "config" in which:
        config_case()
    if "blocks" in which:
        blocks_case(nmodel)
    if "gan_generator" in which:
        gan_generator_case(nmodel)
