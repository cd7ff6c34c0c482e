"""The two-group staggered ring kernel (csrc/gemm_g2.h: 128 x 160 / 128 x 128 tiles, one workgroup per CU) forced on
(NK_GEMM_G2=2) for Linear forward / dgrad / wgrad at full, ragged and K-tail shapes, against fp32 PyTorch on the same
bf16-rounded inputs; plus a race screen: a new synchronisation structure must give bit-identical results run after run with
a second stream loading the chip (cdna guide section 5: "a sync-structure edit makes a NEW template: screen it for races")."""
import os

import pytest
import torch

from tests.util import assert_close, bf16_round

pytestmark = pytest.mark.gpu
TOL_BF16, TOL_F32 = 2e-2, 1e-2


@pytest.fixture(autouse=True)
def force_g2():
    old = os.environ.get("NK_GEMM_G2")
    os.environ["NK_GEMM_G2"] = "2"
    yield
    if old is None:
        os.environ.pop("NK_GEMM_G2", None)
    else:
        os.environ["NK_GEMM_G2"] = old


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return bf16_round(torch.randn(*shape, generator=g) * scale)


def dev(x, dtype=torch.bfloat16):
    return x.to("cuda", dtype=dtype)


# N = 1280 / 640 / 3840 / 160: 160-wide tiles; 1024 / 328: 128-wide ones; ragged M (not a multiple of 128) and K tails (not of 64)
SHAPES = [(4096, 1280, 1280), (16384, 640, 640), (4096, 3840, 1280), (1000, 160, 136), (4000, 1024, 328), (300, 640, 2048), (4096, 1280, 5120), (520, 328, 200)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_g2_linear_fwd(M, N, K):
    from neurosis_amd import ops

    x, w, b, r = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(M, N)
    got = ops.gemm_nt(dev(x), dev(w), dev(b, torch.float32), dev(r))
    assert_close(got, x @ w.t() + b + r, TOL_BF16, "g2 linear_fwd")
    os.environ["NK_GEMM_G2"] = "0"
    old = ops.gemm_nt(dev(x), dev(w), dev(b, torch.float32), dev(r))
    # same k order per output element, same MFMA shape: the two engines should agree to a bf16 ulp or so
    assert float((got.float() - old.float()).abs().max()) <= 2e-2 * float(old.float().abs().max())


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_g2_linear_dgrad(M, N, K):
    """dx[M,K] = dy[M,N] @ w[N,K] (+ dx_add): the output width is K here"""
    from neurosis_amd import ops

    dy, w, a = rnd(M, K), rnd(K, N, scale=K ** -0.5), rnd(M, N)      # output [M, N]: N takes the listed widths
    got = ops.gemm_nn(dev(dy), dev(w), dev(a))
    assert_close(got, dy @ w + a, TOL_BF16, "g2 linear_dgrad")


@pytest.mark.parametrize("M,N,K", [(4096, 5120, 1280), (4096, 10240, 1280), (1000, 2048, 1600), (16384, 5120, 640), (777, 3840, 1280)])
def test_g2_linear_wgrad(M, N, K):
    """dw[N,K] (+)= dy[M,N]^T @ x[M,K]: reduction over M (ragged: K-tail of the GEMM), output width K"""
    from neurosis_amd import ops

    dy, x = rnd(M, N), rnd(M, K)
    ref = dy.t() @ x
    dw = torch.full((N, K), 7.0, device="cuda")
    ops.gemm_tn_f32(dev(dy), dev(x), dw, False)
    assert_close(dw, ref, TOL_F32, "g2 linear_wgrad store")
    ops.gemm_tn_f32(dev(dy), dev(x), dw, True)
    assert_close(dw, 2 * ref, TOL_F32, "g2 linear_wgrad accumulate")


def test_g2_batched_wgrad_three_projections_fill_the_chip():
    """three 1280 x 1280 weight gradients in one launch = 240 tiles of 128 x 160"""
    import ctypes as C

    from neurosis_amd import ops

    M, N, K, n = 4096, 1280, 1280, 3
    dys, xs = [rnd(M, N, seed=i) for i in range(n)], [rnd(M, K, seed=10 + i) for i in range(n)]
    d_dy, d_x = [dev(t) for t in dys], [dev(t) for t in xs]
    dws = [torch.full((N, K), 3.0, device="cuda") for _ in range(n)]
    arr = C.c_void_p * n
    ops.call("nk_linear_wgrad_batched", arr(*[t.data_ptr() for t in d_dy]), arr(*[t.data_ptr() for t in d_x]), arr(*[t.data_ptr() for t in dws]), None, n, M, N, K,
             N, K, K, 0, ops._stream())
    for dy, x, dw in zip(dys, xs, dws):
        assert_close(dw, dy.t() @ x, TOL_F32, "g2 batched wgrad")


def test_g2_race_screen_bitwise_repeatable_under_load():
    from neurosis_amd import ops

    side = torch.cuda.Stream()
    hog_a, hog_b = dev(rnd(8192, 2048, seed=5)), dev(rnd(2048, 2048, seed=6))
    for (M, N, K), kind in [((4096, 1280, 1280), "fwd"), ((4096, 1280, 1280), "dgrad"), ((4096, 1280, 1280), "wgrad"), ((1000, 640, 1096), "fwd"),
                            ((1000, 1024, 1096), "dgrad"), ((3000, 3840, 640), "wgrad")]:
        a, b = dev(rnd(M, K, seed=1)), dev(rnd(N, K, scale=K ** -0.5, seed=2))
        dy = dev(rnd(M, N, seed=3))

        def run():
            if kind == "fwd":
                return ops.gemm_nt(a, b)
            if kind == "dgrad":
                return ops.gemm_nn(dy, b)
            dw = torch.empty(N, K, device="cuda")
            ops.gemm_tn_f32(dy, a, dw, False)
            return dw

        ref = run().clone()
        for it in range(60):
            if it % 2:
                with torch.cuda.stream(side):
                    os.environ["NK_GEMM_G2"] = "0"
                    ops.gemm_nt(hog_a, hog_b)
                    os.environ["NK_GEMM_G2"] = "2"
            out = run()
            assert torch.equal(out, ref), (kind, M, N, K, it)
        torch.cuda.synchronize()


# the UNet's 3 x 3 convolutions on the same kernel (gather A operand; conv-dgrad weights as the transposed (tap, co) x ci operand):
# stride 1 / stride 2 (Downsample) / fused nearest-2x (Upsample), with the ResBlock's row vector and residual, ragged pixel counts
CONV_CASES = [  # N, H, W, Cin, Cout, stride, upsample, rowvec, residual
    (2, 32, 32, 320, 320, 1, False, True, False),
    (2, 16, 16, 640, 1280, 1, False, False, True),
    (1, 24, 40, 1280, 640, 1, False, True, True),        # 960 pixels: 7.5 row tiles
    (2, 32, 32, 320, 320, 2, False, False, False),
    (2, 16, 16, 640, 640, 1, True, False, False),
    (4, 32, 32, 1280, 1280, 1, False, False, False),     # the real 1280-channel level: 256 tiles, one round
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_g2_conv3x3(case):
    from neurosis_amd import ops
    from tests.test_kernels_gpu import _conv_case

    N, H, W, Cin, Cout, stride, up, rowvec, residual = case
    _conv_case(ops, N, H, W, Cin, Cout, 3, stride, 1, upsample=up, rowvec=rowvec, residual=residual)
