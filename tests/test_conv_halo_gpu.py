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


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 128, True), (1, 64, 64, 256, 512, False), (2, 16, 32, 320, 320, True), (1, 30, 62, 128, 256, True)])
def test_halo_fused_groupnorm_prologue_and_statistics_epilogue(case):
    """conv(silu?(GroupNorm(x))) + residual with the GroupNorm applied inside the kernel, and the output's GroupNorm sums from the
    epilogue, against the same three steps as separate launches (GroupNorm kernel -> gather convolution -> statistics kernel)."""
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    N, H, W, Cin, Cout, silu = case
    x = Img(dev(rnd(N * H * W, Cin, scale=1.5) + 0.3), N, H, W)
    norm = torch.nn.GroupNorm(32, Cin, eps=1e-6).cuda()
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.2)
        norm.bias.normal_(0.0, 0.2)
    w = torch.nn.Parameter(ops.conv_weight_param(Cout, Cin, 3, 3).data.normal_(0, (9 * Cin) ** -0.5).cuda(), requires_grad=False)
    bias = torch.randn(Cout, device="cuda")
    res = dev(rnd(N * H * W, Cout))

    os.environ["NK_CONV_HALO"] = "0"
    try:
        xn = ops.groupnorm_fwd(x, norm.weight, norm.bias, 32, norm.eps, silu)[0]
        want = ops.conv2d_fwd(xn, w, bias, residual=res, need_dx=False)[0]
        assert want.sums is None
        want_sums = ops.groupnorm_sums(want, 32)
    finally:
        os.environ.pop("NK_CONV_HALO", None)
    got = ops.conv2d_fwd(x, w, bias, residual=res, need_dx=False, gn=(norm, silu), stats_groups=32)[0]
    assert got.sums is not None, "this shape must take the fused kernel"
    scale = float(want.t.float().abs().max())
    assert float((got.t.float() - want.t.float()).abs().max()) <= 2e-2 * scale
    own = ops.groupnorm_sums(Img(got.t, N, H, W), 32)           # the epilogue's sums are those of the tensor it wrote
    assert float((got.sums - own).abs().max()) <= 1e-4 * float(own.abs().max())
    assert float((got.sums - want_sums).abs().max()) <= 2e-2 * float(want_sums.abs().max())
    # statistics epilogue alone (the UNet's in_layers convolution): same tensor as the plain kernel, bit for bit
    plain = ops.conv2d_fwd(xn, w, bias, residual=res, need_dx=False)[0]
    with_stats = ops.conv2d_fwd(xn, w, bias, residual=res, need_dx=False, stats_groups=32)[0]
    assert torch.equal(plain.t, with_stats.t) and with_stats.sums is not None
    # and GroupNorm consumes the sums instead of its own statistics pass
    a = ops.groupnorm_fwd(with_stats, norm.weight[:Cout] if Cout <= Cin else torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda"), 32, 1e-5, True)[0]
    b = ops.groupnorm_fwd(Img(plain.t, N, H, W), norm.weight[:Cout] if Cout <= Cin else torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda"), 32, 1e-5, True)[0]
    assert float((a.t.float() - b.t.float()).abs().max()) <= 2e-2 * float(b.t.float().abs().max())


def test_vae_encoder_fused_path_equals_unfused_at_real_channel_counts():
    """The frozen SD/SDXL VAE encoder (128..512 channels) on a 128 x 128 image: GroupNorm prologues / statistics epilogues in the
    halo-tile convolutions against the separate-launch path (NK_CONV_HALO=0).  Same tolerances as the encoder's golden test."""
    from neurosis_amd.modules.diffusion.model import Encoder
    from tests.util import cosine, rel_err

    torch.manual_seed(3)
    enc = Encoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0, in_channels=3, resolution=128,
                  z_channels=4, double_z=True, attn_type="vanilla-xformers", standalone=True, embed_dim=4).cuda().requires_grad_(False)
    x = torch.rand(2, 3, 128, 128, device="cuda") * 2 - 1
    got = enc(x, regularize=True)
    os.environ["NK_CONV_HALO"] = "0"
    try:
        want = enc(x, regularize=True)
    finally:
        os.environ.pop("NK_CONV_HALO", None)
    assert got.shape == want.shape == (2, 4, 16, 16)
    assert rel_err(got, want) <= 3e-2 and cosine(got, want) >= 0.999
