"""Frozen CLIP text encoders on the GPU against the fixtures captured from the reference's embedder classes,
plus the kernels they add (causal attention flag, GELU).

Tolerances: bf16 activations through 3-4 transformer layers vs an fp32 CPU path: 3e-2 of the output's max magnitude and
cosine >= 0.999 (as for the UNet); kernels: attention 2e-2, GELU one bf16 ulp of the fp32 formula.
"""
import json
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import synth_state_dict
from tests.util import cosine, rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _fx():
    fx = load_fixture("text_encoders_tiny")
    shapes = json.loads((G / "text_encoders_tiny_keys.json").read_text())
    return fx, synth_state_dict(shapes["hf"]), synth_state_dict(shapes["openclip"])


def _close(got, want, tol=3e-2):
    if isinstance(want, (tuple, list)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            _close(g, w, tol)
    else:
        assert got.shape == want.shape, (got.shape, want.shape)
        assert rel_err(got, want) <= tol, rel_err(got, want)
        assert cosine(got, want) >= 0.999


@pytest.mark.parametrize("B,H,L,D", [(3, 4, 77, 16), (2, 12, 77, 64), (1, 2, 200, 64)])
def test_causal_attention_forward(B, H, L, D):
    from neurosis_amd import ops

    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(B * L, H * D, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    o = ops.attention_fwd(q, k, v, B, H, D, causal=True)[0].float().reshape(B, L, H, D)
    qf, kf, vf = (t.float().reshape(B, L, H, D).transpose(1, 2) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) * D ** -0.5 + torch.full((L, L), float("-inf"), device="cuda").triu(1)
    want = (s.softmax(-1) @ vf).transpose(1, 2)
    assert rel_err(o, want) <= 2e-2
    full = ops.attention_fwd(q, k, v, B, H, D)[0].float().reshape(B, L, H, D)
    assert rel_err(full[:, -1], o[:, -1]) <= 1e-6        # the last query sees every key either way
    assert rel_err(full[:, 0], o[:, 0]) > 1e-2           # the first one does not


def test_causal_attention_has_no_backward():
    from neurosis_amd import ops
    from neurosis_amd.lib import NkError

    q = torch.randn(77, 64).to(torch.bfloat16).cuda()
    o, bwd = ops.attention_fwd(q, q, q, 1, 1, 64, causal=True)
    with pytest.raises(NkError):
        bwd(torch.ones_like(o))


@pytest.mark.parametrize("quick", [False, True])
def test_gelu_kernel(quick):
    from neurosis_amd import ops

    x = torch.linspace(-8, 8, 4096).to(torch.bfloat16).cuda()
    xf = x.float()
    want = xf * torch.sigmoid(1.702 * xf) if quick else torch.nn.functional.gelu(xf)
    got = ops.gelu(x, quick=quick).float()
    assert float((got - want.to(torch.bfloat16).float()).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
    assert rel_err(got, want) <= 5e-3


def test_hf_layout_tower_against_reference():
    from neurosis_amd.models.text_encoder import CLIPTextTower

    fx, hf_sd, _ = _fx()
    tower = CLIPTextTower(**fx["hf_cfg"])
    tower.load_state_dict(hf_sd)                                   # transformers 5.x names (no "text_model." level)
    prefixed = CLIPTextTower(**fx["hf_cfg"])
    prefixed.load_state_dict({"text_model." + k: v for k, v in hf_sd.items()})      # 4.x names, as in SD checkpoints
    assert all(torch.equal(a, b) for a, b in zip(tower.state_dict().values(), prefixed.state_dict().values()))
    tower = tower.cuda()
    out = tower(fx["ids"].cuda(), output_hidden_states=True)
    raw = fx["hf"]["raw"]
    _close(out["last_hidden_state"], raw["last_hidden_state"])
    _close(out["pooler_output"], raw["pooler_output"])
    _close(list(out["hidden_states"]), raw["hidden_states"])


@pytest.mark.parametrize("tag", ["hidden11_style", "penultimate_pooled", "hidden_neg"])
def test_frozen_clip_embedder_against_reference(tag):
    from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder

    fx, hf_sd, _ = _fx()
    case = fx["hf"][tag]
    kw = dict(layer=case["layer"], always_return_pooled=case["return_pooled"], config=fx["hf_cfg"], input_key="caption")
    if case["layer"] == "hidden":
        kw["layer_idx"] = case["layer_idx"]
    emb = FrozenCLIPEmbedder(**kw)
    emb.transformer.load_state_dict(hf_sd)
    emb = emb.cuda()
    assert emb.layer_idx == case["layer_idx"] and not any(p.requires_grad for p in emb.parameters())
    _close(emb(fx["ids"]), case["result"])


@pytest.mark.parametrize("tag", ["penultimate_pooled", "last", "legacy_last", "pooled_layer"])
def test_frozen_openclip_embedder2_against_reference(tag):
    from neurosis_amd.models.text_encoder import FrozenOpenCLIPEmbedder2

    fx, _, oc_sd = _fx()
    case = fx["openclip"][tag]
    emb = FrozenOpenCLIPEmbedder2(layer=case["layer"], always_return_pooled=case["return_pooled"], legacy=case["legacy"], config=fx["openclip_cfg"],
                                  input_key="caption")
    emb.model.load_state_dict(dict(oc_sd, attn_mask=torch.zeros(77, 77)))       # open_clip checkpoints may carry the mask buffer
    emb = emb.cuda()
    _close(emb(fx["ids"]), case["result"])


def test_text_embedders_inside_general_conditioner():
    """the SDXL conditioner layout: CLIP-L hidden states and bigG penultimate states are concatenated on the channel axis
    ("crossattn"), the bigG pooled vector and the size embeddings on "vector"; force_zero_embeddings blanks an embedder."""
    from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2
    from neurosis_amd.modules.encoders import ConcatTimestepEmbedderND, GeneralConditioner

    fx, hf_sd, oc_sd = _fx()
    clip_l = FrozenCLIPEmbedder(layer="hidden", layer_idx=3, config=fx["hf_cfg"], input_key="ids_l")
    clip_l.transformer.load_state_dict(hf_sd)
    big_g = FrozenOpenCLIPEmbedder2(layer="penultimate", always_return_pooled=True, config=fx["openclip_cfg"], input_key="ids_g")
    big_g.model.load_state_dict(oc_sd)
    size = ConcatTimestepEmbedderND(outdim=32, input_key="original_size_as_tuple")
    cond = GeneralConditioner([clip_l, big_g, size]).cuda()
    ids = fx["ids"].cuda()
    batch = {"ids_l": ids, "ids_g": ids, "original_size_as_tuple": torch.tensor([[1024.0, 1024.0]] * 3).cuda()}
    out = cond(batch)
    assert out["crossattn"].shape == (3, 77, 64 + 128) and out["vector"].shape == (3, 96 + 64)
    _close(out["crossattn"][..., :64], fx["hf"]["hidden11_style"]["result"])
    _close(out["crossattn"][..., 64:], fx["openclip"]["penultimate_pooled"]["result"][0])
    _close(out["vector"][:, :96], fx["openclip"]["penultimate_pooled"]["result"][1])
    zero = cond(batch, force_zero_embeddings=["ids_g"])
    assert float(zero["crossattn"][..., 64:].abs().max()) == 0.0 and float(zero["vector"][:, :96].abs().max()) == 0.0
    assert torch.equal(zero["crossattn"][..., :64], out["crossattn"][..., :64])


def test_text_towers_on_two_streams_give_the_in_line_result(monkeypatch):
    """GeneralConditioner runs every second frozen embedder with weights on a side stream (NK_TE_OVERLAP, default on): same kernels on the same
    data, so the outputs equal the in-line order's bit for bit -- repeatedly, with the consumer reading them right behind the join."""
    from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder, FrozenOpenCLIPEmbedder2
    from neurosis_amd.modules.encoders import ConcatTimestepEmbedderND, GeneralConditioner

    fx, hf_sd, oc_sd = _fx()
    clip_l = FrozenCLIPEmbedder(layer="hidden", layer_idx=3, config=fx["hf_cfg"], input_key="ids_l")
    clip_l.transformer.load_state_dict(hf_sd)
    big_g = FrozenOpenCLIPEmbedder2(layer="penultimate", always_return_pooled=True, config=fx["openclip_cfg"], input_key="ids_g")
    big_g.model.load_state_dict(oc_sd)
    cond = GeneralConditioner([clip_l, big_g, ConcatTimestepEmbedderND(outdim=32, input_key="original_size_as_tuple")]).cuda()
    ids = fx["ids"].cuda()
    batch = {"ids_l": ids, "ids_g": ids, "original_size_as_tuple": torch.tensor([[1024.0, 1024.0]] * 3).cuda()}
    monkeypatch.setenv("NK_TE_OVERLAP", "0")
    ref = {k: v.clone() for k, v in cond(batch).items()}
    assert getattr(cond, "_side_stream", None) is None
    monkeypatch.setenv("NK_TE_OVERLAP", "1")
    for _ in range(5):
        got = {k: v.clone() for k, v in cond(batch).items()}          # (read on the consumer stream right behind the join)
        assert all(torch.equal(got[k], ref[k]) for k in ref), {k: float((got[k].float() - ref[k].float()).abs().max()) for k in ref}
    assert cond._side_stream is not None


def test_text_without_a_tokenizer_fails_loudly():
    from neurosis_amd.models.text_encoder import FrozenCLIPEmbedder

    fx, hf_sd, _ = _fx()
    emb = FrozenCLIPEmbedder(layer="penultimate", config=fx["hf_cfg"], input_key="caption").cuda()
    if emb.tokenizer is None:
        with pytest.raises(RuntimeError, match="token ids"):
            emb(["a photo of a cat"])
