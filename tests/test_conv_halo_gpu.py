"""The halo-tile 3 x 3 convolution (csrc/conv_halo.h) against torch's fp32 conv2d on the CPU: the shapes that select it (whole 64-channel
slabs, 128- / 160-multiple output widths, images that 4 x 32 patches cover), with ragged right / bottom tiles, the ResBlock's row vector
and residual, one and several channel slabs, and bit-identity between repeated launches (a race screen for the staggered groups, the
weight ring and the halo double buffer).  Forward, input gradient and weight gradient all go through `ops.conv2d_fwd`, i.e. the C-ABI.
Reference call sites: modules/diffusion/openaimodel.py:247-301, modules/diffusion/model.py:85-134."""
import os

import pytest
import torch

from tests.test_kernels_gpu import _conv_case, dev, rnd

pytestmark = pytest.mark.gpu

HALO_CASES = [  # N, H, W, Cin, Cout, rowvec, residual
    (2, 32, 32, 64, 128, True, True),          # one slab, whole tiles
    (1, 30, 62, 128, 160, True, False),        # ragged right and bottom tiles (30 = 7.5 tile rows, 62 = 1.94 tile columns)
    (2, 30, 62, 64, 320, False, True),         # two column tiles of 160, ragged
    (1, 64, 64, 192, 256, False, False),       # three slabs: the halo double buffer wraps
    (3, 8, 32, 320, 128, True, True),          # five slabs, three images, two tile rows
    (1, 128, 128, 128, 128, False, True),      # the VAE's 128-channel shape at a small size: 512 tiles, two rounds
]


@pytest.mark.parametrize("case", HALO_CASES)
def test_halo_conv3x3(case):
    from neurosis_amd import ops

    N, H, W, Cin, Cout, rowvec, residual = case
    _conv_case(ops, N, H, W, Cin, Cout, 3, 1, 1, rowvec=rowvec, residual=residual)


def test_halo_matches_gather_and_repeats_bitwise():
    """the same convolution through the gather kernels (NK_CONV_HALO=0) agrees to bf16 rounding; 40 repeats of the halo launch with a second
    stream loading the chip are bit-identical"""
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    N, H, W, Cin, Cout = 2, 64, 64, 256, 320
    x = Img(dev(rnd(N * H * W, Cin)), N, H, W)
    w = torch.nn.Parameter(ops.conv_weight_param(Cout, Cin, 3, 3).data.normal_(0, (9 * Cin) ** -0.5).cuda(), requires_grad=False)
    bias = torch.randn(Cout, device="cuda")

    def run():
        return ops.conv2d_fwd(x, w, bias, need_dx=False)[0].t

    os.environ["NK_CONV_HALO"] = "0"
    try:
        gather = run().float()
    finally:
        os.environ.pop("NK_CONV_HALO", None)
    ref = run().clone()
    assert float((ref.float() - gather).abs().max()) <= 2e-2 * float(gather.abs().max())
    side = torch.cuda.Stream()
    hog_a, hog_b = dev(rnd(8192, 1024)), dev(rnd(2048, 1024))
    for it in range(40):
        if it % 2:
            with torch.cuda.stream(side):
                ops.gemm_nt(hog_a, hog_b)
        assert torch.equal(run(), ref), it
    torch.cuda.synchronize()
