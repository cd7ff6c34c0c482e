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
    (3, 96, 96, 128, 320, True, True),         # 432 tiles of 4 x 32: a second, partial round on 256 CUs; two column tiles
    (4, 98, 96, 64, 160, False, True),         # 300 tiles, ONE slab per tile, ragged bottom row
    (5, 64, 64, 256, 128, True, False),        # 320 tiles x 4 slabs, 128-wide column tiles
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


@pytest.mark.parametrize("case", [(2, 32, 64, 128, 128), (1, 64, 64, 256, 512), (2, 16, 32, 320, 320), (1, 30, 62, 128, 256), (4, 32, 32, 1280, 1280),
                                  (3, 96, 96, 128, 320), (4, 98, 96, 64, 160)])       # (the last two: more tiles than CUs)
def test_halo_statistics_epilogue(case):
    """The GroupNorm statistics epilogue: the convolution's output is bit-identical with and without it, the sums it emits are those of
    the tensor it wrote (against the stand-alone statistics pass), and GroupNorm consumes them instead of its own first pass."""
    from neurosis_amd import ops
    from neurosis_amd.ops import Img

    N, H, W, Cin, Cout = case
    x = Img(dev(rnd(N * H * W, Cin, scale=1.5) + 0.3), N, H, W)
    w = torch.nn.Parameter(ops.conv_weight_param(Cout, Cin, 3, 3).data.normal_(0, (9 * Cin) ** -0.5).cuda(), requires_grad=False)
    bias = torch.randn(Cout, device="cuda")
    res = dev(rnd(N * H * W, Cout))
    rowvec = dev(rnd(N, Cout))
    plain = ops.conv2d_fwd(x, w, bias, rowvec=rowvec, residual=res, need_dx=False)[0]
    with_stats = ops.conv2d_fwd(x, w, bias, rowvec=rowvec, residual=res, need_dx=False, stats_groups=32)[0]
    assert plain.sums is None and with_stats.sums is not None, "this shape must take the halo-tile kernel"
    assert torch.equal(plain.t, with_stats.t)
    own = ops.groupnorm_sums(Img(plain.t, N, H, W), 32)
    assert float((with_stats.sums - own).abs().max()) <= 1e-4 * float(own.abs().max())
    gamma, beta = torch.randn(Cout, device="cuda") * 0.2 + 1.0, torch.randn(Cout, device="cuda") * 0.2
    a = ops.groupnorm_fwd(with_stats, gamma, beta, 32, 1e-5, True)[0]           # normalisation pass only
    b = ops.groupnorm_fwd(Img(plain.t, N, H, W), gamma, beta, 32, 1e-5, True)[0]  # statistics pass + normalisation pass
    assert float((a.t.float() - b.t.float()).abs().max()) <= 2e-2 * float(b.t.float().abs().max())


def test_vae_encoder_fused_path_equals_unfused_at_real_channel_counts():
    """The frozen SD/SDXL VAE encoder (128..512 channels) on a 128 x 128 image: halo-tile convolutions with statistics epilogues and
    apply-only GroupNorms against the gather kernels with full GroupNorms (NK_CONV_HALO=0).  Tolerances of the encoder's golden test."""
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


@pytest.mark.parametrize("case", [(2, 3, 128, 40, 56), (1, 4, 320, 33, 17), (2, 3, 8, 16, 16)])
def test_few_channel_conv_in(case):
    """the first convolutions (3 -> 128 of the VAE, 4 -> 320 of the UNet) through nn.Conv2d's channel-padded route: forward on the
    register-resident FMA kernel, weight / bias gradients on the tile engine as before; against torch's fp32 conv2d"""
    import torch.nn.functional as F

    from neurosis_amd import nn as nkn, ops
    from neurosis_amd.ops import Img
    from tests.util import assert_close

    N, Cin, Cout, H, W = case
    torch.manual_seed(7)
    conv = nkn.Conv2d(Cin, Cout, 3, padding=1).cuda()
    x = rnd(N, Cin, H, W)
    xt = ops.nchw_to_tokens(dev(x, torch.float32), 8)
    out, bwd = conv.fwd(Img(xt, N, H, W), need_dx=False)
    wq = conv.weight.detach().float().cpu().to(torch.bfloat16).float().requires_grad_(True)
    bq = conv.bias.detach().float().cpu().requires_grad_(True)
    ref = F.conv2d(x, wq, bq, padding=1)
    pad = (Cout + 7) // 8 * 8
    got = out.t.view(N, H, W, pad).permute(0, 3, 1, 2)[:, :Cout]
    assert_close(got, ref, 2e-2, "few-channel conv fwd")
    dy = rnd(N, Cout, H, W)
    ref.backward(dy)
    dyt = torch.zeros(N, H, W, pad)
    dyt[..., :Cout] = dy.permute(0, 2, 3, 1)
    bwd(dev(dyt.reshape(-1, pad)))
    ops.join_wgrad_stream()
    assert_close(conv.weight.grad, wq.grad, 1e-2, "few-channel conv wgrad")
    assert_close(conv.bias.grad, bq.grad, 1e-2, "few-channel conv bias grad")
