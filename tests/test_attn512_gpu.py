"""Head-dim-512 flash attention forward (csrc/attn512.h): the VAE mid block's single head,
reference modules/diffusion/model.py:224-243 (softmax(q k^T / sqrt(C)) v over H*W tokens)."""
import os
import re
import shutil
import subprocess

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "neurosis_amd", "csrc")


def _ref(q, k, v, B):
    L, D = q.shape[0] // B, q.shape[1]
    Lk = k.shape[0] // B
    qf, kf, vf = q.float().view(B, L, D), k.float().view(B, Lk, D), v.float().view(B, Lk, D)
    s = torch.einsum("bqd,bkd->bqk", qf, kf) * D ** -0.5
    return torch.einsum("bqk,bkd->bqd", torch.softmax(s, -1), vf).reshape(B * L, D), torch.logsumexp(s, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,Lq,Lk", [(2, 256, 256), (1, 128, 32), (3, 200, 200), (1, 96, 333), (2, 1024, 1024)])
def test_attn512_forward_matches_fp32_softmax(B, Lq, Lk):
    from neurosis_amd import ops

    torch.manual_seed(B * 1000 + Lq + Lk)
    dev = "cuda"
    q = (torch.randn(B * Lq, 512, device=dev) * 1.5).to(torch.bfloat16)
    k = (torch.randn(B * Lk, 512, device=dev) * 1.5).to(torch.bfloat16)
    v = torch.randn(B * Lk, 512, device=dev).to(torch.bfloat16)
    o, _ = ops.attention_fwd(q, k, v, B, 1, 512)
    ref, lse = _ref(q, k, v, B)
    err = (o.float() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err          # bf16 output, bf16 probabilities: the tolerance of every attention test
    cos = torch.nn.functional.cosine_similarity(o.float().flatten(), ref.flatten(), dim=0).item()
    assert cos >= 0.9995, cos


@pytest.mark.gpu
def test_attn512_lse_and_rising_maximum():
    """Keys ordered so that the running maximum rises by more than the deferred-rescale threshold (2^8) several times:
    exercises the AGPR rescale path; the log-sum-exp output pins m and l."""
    import ctypes as C

    from neurosis_amd import ops
    from neurosis_amd.lib import NkAttnDesc

    torch.manual_seed(7)
    dev = "cuda"
    B, L = 1, 512
    q = torch.randn(B * L, 512, device=dev).to(torch.bfloat16)
    k = torch.randn(B * L, 512, device=dev)
    k = (k * torch.linspace(0.05, 6.0, L, device=dev)[:, None]).to(torch.bfloat16)      # later keys score far higher (and far lower)
    v = torch.randn(B * L, 512, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    lse = torch.empty(B, 1, L, dtype=torch.float32, device=dev)
    d = NkAttnDesc()
    d.B, d.H, d.Lq, d.Lk, d.D = B, 1, L, L, 512
    d.sq = d.sk = d.sv = d.so = 512
    d.bq = d.bk = d.bv = d.bo = L * 512
    d.scale = 512 ** -0.5
    d.causal = 0
    ops.call("nk_attention_fwd", C.byref(d), q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), lse.data_ptr(), ops._stream())
    ref, ref_lse = _ref(q, k, v, B)
    assert (ref_lse.max() - ref_lse.min()).item() > 10.0           # the case is what it claims to be
    # q * scale is rounded to bf16 once (2^-9 relative) and these scores reach ~20: tolerance relative to the magnitude
    assert (lse.view(B, L) - ref_lse).abs().max().item() <= 3e-3 * ref_lse.abs().max().item()
    assert (o.float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
def test_attn512_full_size_agrees_with_the_two_gemm_path():
    """BASELINE config 2's shape, one image: 128 x 128 latent tokens.  The unfused path (q k^T GEMM, row softmax, p v GEMM) is the
    independent implementation; both round probabilities to bf16."""
    from neurosis_amd import ops

    torch.manual_seed(3)
    dev = "cuda"
    L = 16384
    q = torch.randn(L, 512, device=dev).to(torch.bfloat16)
    k = torch.randn(L, 512, device=dev).to(torch.bfloat16)
    v = torch.randn(L, 512, device=dev).to(torch.bfloat16)
    o = ops.attention_fwd(q, k, v, 1, 1, 512, need_lse=False)[0]
    o2 = ops.attention_unfused(q, k, v, 1)
    assert torch.isfinite(o.float()).all()
    err = (o.float() - o2.float()).abs().max().item()
    assert err <= 2e-2 * max(1.0, o2.float().abs().max().item()), err
    rows = torch.randint(0, L, (64,), device=dev)
    s = (q[rows].float() @ k.float().t()) * 512 ** -0.5
    ref = torch.softmax(s, -1) @ v.float()
    assert (o[rows].float() - ref).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_attn512_accumulator_file_is_not_shared_with_the_compiler():
    """attn512.h addresses the AGPR file physically from inline asm.  That is only sound while hipcc allocates no AGPR of its own in
    the kernel and spills nothing inside the key loop: check the ISA it generates (no GPU needed)."""
    from tools.check_attn512_isa import check

    check(CSRC)
