"""Fused multi-tensor Adafactor on the flat buffers (csrc/optim.hip) against the reference's own steps (golden fixture)
and against the CPU oracle on a larger, SDXL-like parameter set.  fp32 throughout: tolerance 2e-5 relative on parameters
(reduction orders differ), 1e-4 on the factored states."""
from pathlib import Path

import pytest
import torch

from tests.util import rel_err
from tests.golden.fixture_io import load_fixture

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


def make_store(tensors):
    from neurosis_amd.nn import FlatParamStore

    # conv weights live channels-last (the store's physical layout is [O][KH][KW][I])
    params = [torch.nn.Parameter(t.clone().cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.clone().cuda()) for t in tensors]
    return FlatParamStore(params), params


def set_grads(params, grads, scale=1.0):
    for p, g in zip(params, grads):
        p.grad.copy_(g.cuda() * scale)


@pytest.mark.parametrize("streams", ["1", "2"])
@pytest.mark.parametrize("tag", ["relative", "manual"])
def test_flat_adafactor_matches_reference_steps(tag, streams, monkeypatch):
    """(streams = 2: NK_AF_STREAMS=2, consecutive chunks alternating between two streams with a workspace each)"""
    from neurosis_amd.optim import FlatAdafactor

    monkeypatch.setenv("NK_AF_STREAMS", streams)
    c = load_fixture("adafactor_steps")[tag]
    store, params = make_store(c["init"])
    opt = FlatAdafactor(store, chunk_bytes=8 << 10, **c["kwargs"])   # tiny chunks: several chunks even for this small set
    assert len(opt.chunks) > 2
    for s in range(3):
        set_grads(params, c["grads"][s])
        opt.step()
        torch.cuda.synchronize()
        for p, want in zip(params, c["after"][s]):
            assert rel_err(p.detach().cpu(), want) <= 2e-5, (s, tuple(p.shape))
        # the bf16 shadows the kernels read follow the masters
        for p in params:
            from neurosis_amd import ops

            assert rel_err(ops._phys_flat(p).float().cpu(), p._nk_shadow.float().cpu()) <= 1e-2
    for i, want in enumerate(c["states"]):
        got = opt.param_state(i)
        for k in ("exp_avg_sq_row", "exp_avg_sq_col", "exp_avg_sq"):
            if k in want:
                assert rel_err(got[k].cpu(), want[k]) <= 1e-4, (i, k)
    # grad_scale (the data-parallel mean): scaled gradients + grad_scale == unscaled gradients
    store2, params2 = make_store(c["init"])
    opt2 = FlatAdafactor(store2, **c["kwargs"])
    set_grads(params2, c["grads"][0], scale=4.0)
    opt2.step(grad_scale=0.25)
    for p, want in zip(params2, c["after"][0]):
        assert rel_err(p.detach().cpu(), want) <= 2e-5


def test_flat_adafactor_sdxl_like_shapes_vs_oracle():
    """Real channel counts (ragged tiles: 1280x2048 context projections, 320-channel convs, 4-channel conv_out, 2816 label
    input), five steps, against the CPU oracle."""
    from neurosis_amd.optim import AdafactorScheduler, FlatAdafactor
    from oracle import adafactor_oracle as AO

    g = torch.Generator().manual_seed(7)
    shapes = [(1280, 2048), (2560, 640), (640, 2560), (1280, 2816), (330, 1284), (320, 320, 3, 3), (640, 320, 1, 1), (8, 320, 3, 3),
              (320, 8, 3, 3), (1280,), (5000,), (3,)]
    init = [torch.randn(*s, generator=g) * 0.05 for s in shapes]
    store, params = make_store(init)
    kw = dict(scale_parameter=True, relative_step=True, warmup_init=True)
    opt = FlatAdafactor(store, **kw)
    sched = AdafactorScheduler(opt, initial_lr=4e-7)
    assert sched.get_lr() == [4e-7]
    ref = [t.clone() for t in init]
    sts = [AO.new_state(p) for p in ref]
    for s in range(5):
        grads = [torch.randn(*sh, generator=g) * (0.1 if s % 2 else 3.0) for sh in shapes]
        set_grads(params, grads)
        opt.step()
        lrs = [AO.step_tensor(p, gr, st, **kw) for p, gr, st in zip(ref, grads, sts)]
        torch.cuda.synchronize()
        for p, want in zip(params, ref):
            assert rel_err(p.detach().cpu(), want) <= 2e-5, (s, tuple(p.shape))
        # lr = max(eps2, RMS(p)) * rel_step: torch's CPU norm over 2-4 M elements carries ~1e-5..1e-4 of fp32 accumulation noise
        assert rel_err(opt.current_lrs().cpu(), torch.tensor(lrs)) <= 3e-4
    assert abs(sched.get_lr()[0] - lrs[0]) <= 3e-4 * lrs[0]


def test_flat_adafactor_argument_errors():
    from neurosis_amd.optim import FlatAdafactor

    store, _ = make_store([torch.zeros(8, 8)])
    with pytest.raises(ValueError):
        FlatAdafactor(store, lr=1e-3, relative_step=True)
    with pytest.raises(ValueError):
        FlatAdafactor(store, lr=1e-3, relative_step=False, warmup_init=True)
    with pytest.raises(NotImplementedError):
        FlatAdafactor(store, beta1=0.9)


def test_flat_ema_matches_lit_ema_rule():
    """modules/ema.py:40-59: shadow -= (1 - d) * (shadow - p), d = min(decay, (1+n)/(10+n)); store / copy_to / restore."""
    from neurosis_amd.optim import FlatEma

    g = torch.Generator().manual_seed(3)
    init = [torch.randn(40, 24, generator=g), torch.randn(8, 8, 3, 3, generator=g), torch.randn(17, generator=g)]
    store, params = make_store(init)
    ema = FlatEma(store, decay=0.999)
    ref = [t.clone() for t in init]
    cur = [t.clone() for t in init]
    for n in range(1, 4):
        cur = [c + 0.1 * torch.randn(c.shape, generator=g) for c in cur]
        with torch.no_grad():
            for p, c in zip(params, cur):
                p.copy_(c.cuda())
        ema.update()
        d = min(0.999, (1 + n) / (10 + n))
        ref = [r - (1.0 - d) * (r - c) for r, c in zip(ref, cur)]
    torch.cuda.synchronize()
    ema.store()
    ema.copy_to()
    for p, want in zip(params, ref):
        assert rel_err(p.detach().cpu(), want) <= 1e-6
    ema.restore()
    for p, want in zip(params, cur):
        assert rel_err(p.detach().cpu(), want) <= 1e-6
