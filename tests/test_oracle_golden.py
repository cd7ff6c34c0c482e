"""The CPU oracle (oracle/sdxl_oracle.py) against golden vectors captured from the reference itself
(tests/golden/make_golden.py).  fp32 CPU restatement vs fp32 CPU reference: tolerance 1e-5 relative
(SURVEY section 8(c)); observed differences are at the 1e-6 level (different but equivalent op order)."""
import json
from pathlib import Path

import pytest
import torch

from oracle import sdxl_oracle as O
from tests.golden.make_golden import synth_state_dict
from tests.util import rel_err

G = Path(__file__).resolve().parent / "golden"
TOL = 1e-5


def _load(name):
    fx = torch.load(G / f"{name}.pt", weights_only=False)
    shapes = json.loads((G / f"{name}_keys.json").read_text())
    return fx, synth_state_dict(shapes)


@pytest.mark.parametrize("name", ["unet_sdxl_tiny", "unet_sd15_tiny"])
def test_unet_forward_loss_and_grads(name):
    fx, sd = _load(name)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cfg = fx["cfg"]
    table = O.legacy_ddpm_sigmas()
    assert torch.equal(table, fx["sigma_table"])
    idx = O.sigma_to_idx(table, table[O.sigma_to_idx(table, fx["sigma"])])
    assert torch.equal(idx, fx["c_noise_idx"])
    z_t = fx["x"] + fx["sigma"][:, None, None, None] * fx["noise"]
    assert rel_err(z_t, fx["z_t"]) < 1e-6
    with torch.no_grad():
        c_in = (1.0 / (table[idx] ** 2 + 1.0) ** 0.5)[:, None, None, None]
        f_out = O.unet_forward(sd, cfg, z_t * c_in, idx, fx["context"], fx["y"])
    assert rel_err(f_out, fx["F_out"]) < TOL

    def net(xin, t):
        return O.unet_forward(sd, cfg, xin, t, fx["context"], fx["y"])

    loss = O.edm_loss(net, table, fx["x"], fx["sigma"], fx["noise"])
    assert rel_err(loss, fx["loss"]) < TOL
    loss.mean().backward()
    for k, g in fx["grads"].items():
        assert rel_err(sd[k].grad, g) < 1e-4, k
    for k, n in fx["grad_norms"].items():
        assert abs(float(sd[k].grad.norm()) - n) <= 1e-4 * max(n, 1e-6) + 1e-7, k


def test_vae_encoder():
    fx, sd = _load("vae_encoder_tiny")
    z = O.vae_encode(sd, fx["cfg"], fx["image"])
    assert rel_err(z, fx["z"]) < TOL
    assert rel_err(z, fx["moments"][:, :4]) < TOL


def test_glue_vectors():
    fx = torch.load(G / "glue_vectors.pt", weights_only=False)
    assert rel_err(O.timestep_embedding(fx["t"], 320), fx["emb320"]) < 1e-6
    assert torch.equal(O.legacy_ddpm_sigmas(), fx["ddpm_table"])
