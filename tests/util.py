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
    # fp64: an fp32 dot / norm over 10^7 elements is itself off by ~1e-3 (the full-depth SDXL test read cosines of 1.0019)
    got = got.detach().cpu().reshape(-1).double()
    ref = ref.detach().cpu().reshape(-1).double()
    return float(torch.dot(got, ref) / (got.norm() * ref.norm()).clamp_min(1e-30))


def assert_close(got, ref, tol, name=""):
    e = rel_err(got, ref)
    c = cosine(got, ref)
    assert torch.isfinite(got.detach().float()).all(), f"{name}: non-finite output"
    assert e <= tol, f"{name}: normalised max error {e:.3e} > {tol:.1e} (cos={c:.6f})"
    assert c >= 1.0 - 10 * tol * tol - 1e-4, f"{name}: cosine {c:.6f} too low"


WORST_COSINES = {}     # test label -> (worst cosine over >= 2-D parameters, its name, worst over 1-D parameters, its name): printed by the tests


def check_grad_cosines(label, named_grads, refs, floor_matrix=0.999, floor_vector=0.99, keep=None):
    """Gradient direction against the reference, parameter by parameter: weight matrices / convolution kernels (>= 2-D) must reach
    `floor_matrix`, 1-D parameters (biases, norm scales: sums of 10^5..10^7 bf16-rounded products, often nearly cancelling) `floor_vector`.
    `keep(name, ref)` filters (e.g. analytically-zero gradients).  Records and prints the worst observed value of each tier."""
    worst = {True: (1.0, None), False: (1.0, None)}
    failures = []
    for k, g in refs.items():
        if keep is not None and not keep(k, g):
            continue
        got = named_grads[k]
        got = got.grad if getattr(got, "grad", None) is not None and got.grad.shape == g.shape else got
        c = cosine(got, g)
        matrix = g.dim() >= 2
        if c < worst[matrix][0]:
            worst[matrix] = (c, k)
        if c < (floor_matrix if matrix else floor_vector):
            failures.append((k, tuple(g.shape), round(c, 6)))
    WORST_COSINES[label] = worst[True] + worst[False]
    print(f"[gradient cosines] {label}: worst >=2-D {worst[True][0]:.6f} ({worst[True][1]}), worst 1-D {worst[False][0]:.6f} ({worst[False][1]})")
    assert not failures, (label, failures[:8])
    return worst
