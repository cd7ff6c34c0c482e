"""oracle/clip_oracle.py against the fixtures captured from the reference's embedder classes (tests/golden/make_golden.py::
text_encoder_cases).  fp32 vs fp32: 1e-5 relative."""
import json
from pathlib import Path

import pytest
import torch

from oracle import clip_oracle as CO
from tests.golden.make_golden import synth_state_dict
from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"
TOL = 1e-5


def _fx():
    fx = load_fixture("text_encoders_tiny")
    shapes = json.loads((G / "text_encoders_tiny_keys.json").read_text())
    return fx, synth_state_dict(shapes["hf"]), synth_state_dict(shapes["openclip"])


def _close(got, want):
    if isinstance(want, (tuple, list)):
        assert len(got) == len(want)
        for g, w in zip(got, want):
            _close(g, w)
    else:
        assert got.shape == want.shape and rel_err(got, want) < TOL, rel_err(got, want)


def test_hf_clip_text_model_oracle():
    fx, hf_sd, _ = _fx()
    out = CO.hf_text_model(hf_sd, fx["hf_cfg"], fx["ids"])
    raw = fx["hf"]["raw"]
    _close(out["last_hidden_state"], raw["last_hidden_state"])
    _close(out["pooler_output"], raw["pooler_output"])
    _close(out["hidden_states"], raw["hidden_states"])
    for tag, case in fx["hf"].items():
        if tag != "raw":
            _close(CO.frozen_clip_embedder(hf_sd, fx["hf_cfg"], fx["ids"], case["layer"], case["layer_idx"], case["return_pooled"]), case["result"])


def test_openclip_text_tower_oracle():
    fx, _, oc_sd = _fx()
    for tag, case in fx["openclip"].items():
        _close(CO.frozen_openclip_embedder2(oc_sd, fx["openclip_cfg"], fx["ids"], case["layer"], case["return_pooled"], case["legacy"]), case["result"])


def test_causal_mask_matters_and_padding_is_ignored_by_the_pooled_row():
    """sanity on the fixture itself: changing a token changes only later positions; the pooled row is the first EOS"""
    fx, hf_sd, _ = _fx()
    ids = fx["ids"].clone()
    base = CO.hf_text_model(hf_sd, fx["hf_cfg"], ids)["last_hidden_state"]
    ids[0, 5] = 17
    moved = CO.hf_text_model(hf_sd, fx["hf_cfg"], ids)["last_hidden_state"]
    assert torch.equal(moved[0, :5], base[0, :5]) and not torch.equal(moved[0, 5:], base[0, 5:])
    assert torch.equal(moved[1:], base[1:])
