"""oracle/lpips_oracle.py against the reference's LPIPS.forward (tests/golden/make_golden.py::lpips_case).  fp32: 1e-5."""
import json
from pathlib import Path

import torch

from oracle import lpips_oracle as LO
from tests.golden.make_golden import synth_state_dict
from tests.util import rel_err

G = Path(__file__).resolve().parent / "golden"


def trunk_weights():
    shapes = json.loads((G / "lpips_vgg_tiny_keys.json").read_text())
    return {k: v * 1.6 for k, v in synth_state_dict(shapes).items()}


def test_lpips_oracle_distance_and_gradient():
    fx = torch.load(G / "lpips_vgg_tiny.pt", weights_only=False)
    y = fx["y"].clone().requires_grad_(True)
    dist = LO.lpips(trunk_weights(), fx["lin"], fx["x"], y)
    assert dist.shape == fx["distance"].shape and rel_err(dist, fx["distance"]) < 1e-5
    (dist.reshape(-1) * fx["upstream"]).sum().backward()
    assert rel_err(y.grad, fx["d_y"]) < 1e-4
    assert all(float(v.min()) >= 0 for v in fx["lin"].values())       # the calibrated lin weights are non-negative
