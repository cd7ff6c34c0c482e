"""oracle/lpips_oracle.py against the reference's LPIPS.forward (tests/golden/make_golden.py::lpips_case).  fp32: 1e-5."""
import json
from pathlib import Path

import pytest
import torch

from oracle import lpips_oracle as LO
from tests.golden.make_golden import synth_state_dict
from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"


def trunk_weights(kind="vgg"):
    shapes = json.loads((G / f"lpips_{kind}_tiny_keys.json").read_text())
    return {k: v * 1.6 for k, v in synth_state_dict(shapes).items()}


@pytest.mark.parametrize("kind", ["vgg", "alex"])
def test_lpips_oracle_distance_and_gradient(kind):
    fx = load_fixture(f"lpips_{kind}_tiny")
    y = fx["y"].clone().requires_grad_(True)
    dist = LO.lpips(trunk_weights(kind), fx["lin"], fx["x"], y, trunk=kind)
    assert dist.shape == fx["distance"].shape and rel_err(dist, fx["distance"]) < 1e-5
    (dist.reshape(-1) * fx["upstream"]).sum().backward()
    assert rel_err(y.grad, fx["d_y"]) < 1e-4
    assert all(float(v.min()) >= 0 for v in fx["lin"].values())       # the calibrated lin weights are non-negative
