"""torch.library surface (neurosis_amd.torch_ops, namespace `neurosis_hip`): every op is registered with a schema, has a Meta
kernel with the right shapes / dtypes (FakeTensor tracing), and has NO CPU kernel -- the product never falls back."""
import pytest
import torch

import neurosis_amd.torch_ops as T

bf = torch.bfloat16


def test_every_op_is_registered_with_a_schema():
    for name in T.OPS:
        op = getattr(torch.ops.neurosis_hip, name)
        assert str(op.default._schema).startswith(f"neurosis_hip::{name}("), name


def test_meta_kernels_give_the_right_shapes():
    m = lambda *s, dtype=bf: torch.empty(*s, dtype=dtype, device="meta")
    o = torch.ops.neurosis_hip
    assert o.linear(m(256, 64), m(96, 64), m(96, dtype=torch.float32)).shape == (256, 96)
    assert o.linear_wgrad(m(256, 96), m(256, 64)).shape == (96, 64) and o.linear_wgrad(m(256, 96), m(256, 64)).dtype == torch.float32
    y, mean, rstd = o.layernorm_fwd(m(256, 64), m(64, dtype=torch.float32), m(64, dtype=torch.float32), 1e-5)
    assert y.shape == (256, 64) and mean.shape == (256,) and rstd.dtype == torch.float32
    y, mean, rstd = o.groupnorm_silu_fwd(m(2 * 64, 64), m(64, dtype=torch.float32), m(64, dtype=torch.float32), 2, 32, 1e-5, True)
    assert y.shape == (128, 64) and mean.shape == (2, 32)
    assert o.geglu(m(256, 128)).shape == (256, 64)
    out, lse = o.attention_fwd(m(2 * 128, 4 * 64), m(2 * 77, 4 * 64), m(2 * 77, 4 * 64), 2, 4)
    assert out.shape == (256, 256) and lse.shape == (2, 4, 128)
    assert o.conv2d(m(2 * 16 * 16, 32), m(64, 3, 3, 32), None, 2, 16, 16, 2, 1).shape == (2 * 8 * 8, 64)
    assert o.conv2d_wgrad(m(2 * 8 * 8, 64), m(2 * 16 * 16, 32), 64, 3, 3, 2, 16, 16, 2, 1).shape == (64, 3, 3, 32)
    assert o.timestep_embedding(m(4, dtype=torch.float32), 320).shape == (4, 320)
    # the families added in round 3 (SURVEY 8(b)'s list: upsample2x_nearest_cat, edm_loss_{fwd,bwd}, flat_allreduce_{start,wait})
    assert o.cat_channels(m(128, 32), m(128, 64)).shape == (128, 96)
    a, b = o.split_channels(m(128, 96), 32)
    assert a.shape == (128, 32) and b.shape == (128, 64)
    assert o.upsample2x_nearest_bwd(m(2 * 16 * 16, 64), 2, 8, 8).shape == (2 * 8 * 8, 64)
    assert o.upsample2x_nearest_conv(m(2 * 8 * 8, 64), m(128, 3, 3, 64), None, 2, 8, 8).shape == (2 * 16 * 16, 128)
    f32 = torch.float32
    zt, net_in = o.edm_prepare(m(2, 4, 16, 16, dtype=f32), m(2, 4, 16, 16, dtype=f32), m(2, dtype=f32), m(2, dtype=f32), 8)
    assert zt.shape == (2, 4, 16, 16) and net_in.shape == (512, 8) and net_in.dtype == bf
    assert o.edm_loss(m(512, 8), m(2, 4, 16, 16, dtype=f32), m(2, 4, 16, 16, dtype=f32), m(2, dtype=f32), m(2, dtype=f32), m(2, dtype=f32)).shape == (2,)
    assert o.edm_loss_bwd(m(2, dtype=f32), m(512, 8), m(2, 4, 16, 16, dtype=f32), m(2, 4, 16, 16, dtype=f32), m(2, dtype=f32), m(2, dtype=f32), m(2, dtype=f32)).shape == (512, 8)
    assert o.flat_allreduce_start(m(1000, dtype=f32), 0, 500) is None and o.flat_allreduce_wait(m(1000, dtype=f32)) is None
    y, sums = o.conv2d_fwd_stats(m(2 * 32 * 32, 64), m(128, 3, 3, 64), None, 2, 32, 32, 32)
    assert y.shape == (2048, 128) and sums.shape == (2, 64) and sums.dtype == f32
    assert o.linear_dgrad_geglu(m(256, 64), m(64, 128), m(256, 256)).shape == (256, 256)
    u, h = o.linear_fwd_geglu(m(256, 64), m(512, 64), None)
    assert u.shape == (256, 512) and h.shape == (256, 256)
    assert o.nchw_to_nlc(m(2, 4, 8, 8, dtype=torch.float32), 8).shape == (128, 8)
    assert o.nlc_to_nchw(m(128, 8), 2, 4, 8, 8).shape == (2, 4, 8, 8)


def test_there_is_no_cpu_kernel():
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.neurosis_hip.linear(torch.zeros(8, 8, dtype=bf), torch.zeros(8, 8, dtype=bf), None)
