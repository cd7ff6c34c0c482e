"""Shared helpers for the parity tests."""
import torch


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 values exactly representable in bf16 (so the HIP path and the fp32 oracle see identical inputs)."""
    return x.to(torch.bfloat16).to(torch.float32)


def rel_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    """max |got-ref| normalised by max |ref| (robust for bf16 outputs whose small entries carry absolute noise)."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    denom = ref.abs().max().clamp_min(1e-12)
    return float((got - ref).abs().max() / denom)


def cosine(got: torch.Tensor, ref: torch.Tensor) -> float:
    got = got.detach().float().cpu().reshape(-1)
    ref = ref.detach().float().cpu().reshape(-1)
    return float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-30))


def assert_close(got, ref, tol, name=""):
    e = rel_err(got, ref)
    c = cosine(got, ref)
    assert torch.isfinite(got.detach().float()).all(), f"{name}: non-finite output"
    assert e <= tol, f"{name}: normalised max error {e:.3e} > {tol:.1e} (cos={c:.6f})"
    assert c >= 1.0 - 10 * tol * tol - 1e-4, f"{name}: cosine {c:.6f} too low"
