"""The halo-tile weight gradient of 3 x 3 / stride-1 / padding-1 convolutions (csrc/conv_wgrad_halo.h) through the C-ABI
(nk_conv2d_wgrad / nk_conv2d_wgrad_bias): against torch's fp32 conv2d autograd on the CPU and against the implicit-GEMM gather kernel it
replaces (NK_CONV_WGRAD_HALO=0), over the things the kernel has to get right -- several (co, ci) blocks, a last co block that is partly
empty (320 = 2.5 x 128), ragged right / bottom pixel tiles, pixel ranges split over workgroups (atomics into a zeroed destination) and not
split (plain stores), the add-to-what-is-there mode of gradient accumulation, the fused bias gradient.  The forward / dgrad / wgrad cases of
tests/test_conv_halo_gpu.py run through the same kernel as well.  Reference: autograd of the convolutions at
modules/diffusion/openaimodel.py:247-301 and modules/diffusion/model.py:85-134."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import assert_close, dev, rnd

pytestmark = pytest.mark.gpu

CASES = [  # N, H, W, Cin, Cout
    (2, 8, 32, 64, 128),         # one block, four pixel tiles: split over workgroups
    (1, 16, 64, 128, 320),       # 3 co blocks (the last half empty) x 2 ci blocks
    (2, 30, 62, 64, 160),        # ragged right and bottom tiles, second co block a quarter full
    (1, 10, 64, 192, 64),        # ragged bottom, half a co block, three ci blocks
    (4, 64, 64, 64, 64),         # 128 pixel tiles into ONE block: many splits, atomics
    (2, 32, 32, 1280, 1280),     # 200 blocks (the UNet's lowest level at half batch): one pixel range per block, plain stores
]


@pytest.fixture(autouse=True)
def _every_eligible_shape(monkeypatch):
    monkeypatch.setenv("NK_CONV_WGRAD_HALO", "2")       # (the by-shape rule would send the small cases to the gather kernel)


def _run(ops, x, dy, N, H, W, Cin, Cout, accumulate, dw=None, db=None, bias=True):
    from neurosis_amd.lib import call

    d = ops._conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, H, W, False)
    if dw is None:
        dw = torch.full((Cout, 9 * Cin), 7.0, device="cuda")          # (overwrite mode must not care what was there)
        db = torch.full((Cout,), 7.0, device="cuda")
    if bias:
        call("nk_conv2d_wgrad_bias", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), accumulate, ops._stream())
    else:
        call("nk_conv2d_wgrad", C.byref(d), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), accumulate, ops._stream())
    torch.cuda.synchronize()
    return dw, db


@pytest.mark.parametrize("case", CASES)
def test_wgrad_halo_against_autograd_and_the_gather_kernel(case):
    from neurosis_amd import ops

    N, H, W, Cin, Cout = case
    x = rnd(N, Cin, H, W)
    dy = rnd(N, Cout, H, W)
    xq, dyq = x.bfloat16().float(), dy.bfloat16().float()
    w = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(xq, w, padding=1).backward(dyq)
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)            # [co][tap][ci]: the physical layout of the weights
    xd = dev(x).permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()
    dyd = dev(dy).permute(0, 2, 3, 1).reshape(-1, Cout).contiguous()

    dw, db = _run(ops, xd, dyd, N, H, W, Cin, Cout, 0)
    assert_close(dw, ref, 1e-2, "halo wgrad")
    assert_close(db, dyq.sum((0, 2, 3)), 1e-2, "halo wgrad bias gradient")
    # add mode: twice the gradient on top of what the first launch left
    dw2, db2 = _run(ops, xd, dyd, N, H, W, Cin, Cout, 1, dw.clone(), db.clone())
    assert_close(dw2, 2 * ref, 1e-2, "halo wgrad accumulate")
    assert_close(db2, 2 * dyq.sum((0, 2, 3)), 1e-2, "halo wgrad bias accumulate")
    # destination known to be zero (the flat gradient buffer right after zero_grad)
    dw3, _ = _run(ops, xd, dyd, N, H, W, Cin, Cout, 2, torch.zeros_like(dw), torch.zeros_like(db))
    assert_close(dw3, ref, 1e-2, "halo wgrad into a zeroed destination")
    # without the bias gradient
    dw4, _ = _run(ops, xd, dyd, N, H, W, Cin, Cout, 0, bias=False)
    assert torch.equal(dw4, dw) or float((dw4 - dw).abs().max()) <= 1e-4 * float(ref.abs().max())      # (atomics order only)
    # the kernel it replaces
    os.environ["NK_CONV_WGRAD_HALO"] = "0"
    try:
        dwg, dbg = _run(ops, xd, dyd, N, H, W, Cin, Cout, 0)
    finally:
        os.environ["NK_CONV_WGRAD_HALO"] = "2"
    scale = float(ref.abs().max())
    assert float((dwg - dw).abs().max()) <= 2e-4 * scale, "halo vs gather (both accumulate in fp32; only the order differs)"
    assert float((dbg - db).abs().max()) <= 2e-4 * float(db.abs().max())


def test_wgrad_halo_repeats_agree_under_load():
    """A race screen for the double-buffered stages: 30 launches with a second stream loading the chip; unsplit launches (one pixel range per
    block, plain stores) must be bit-identical."""
    from neurosis_amd import ops

    N, H, W, Cin, Cout = 2, 32, 32, 1280, 640
    xd = dev(rnd(N * H * W, Cin))
    dyd = dev(rnd(N * H * W, Cout))
    ref, _ = _run(ops, xd, dyd, N, H, W, Cin, Cout, 0, bias=False)
    ref = ref.clone()
    side = torch.cuda.Stream()
    a, b = dev(rnd(8192, 1024)), dev(rnd(2048, 1024))
    for it in range(30):
        if it % 2:
            with torch.cuda.stream(side):
                ops.gemm_nt(a, b)
        got, _ = _run(ops, xd, dyd, N, H, W, Cin, Cout, 0, bias=False)
        assert torch.equal(got, ref), it
