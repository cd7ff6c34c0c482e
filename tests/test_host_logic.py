"""Host-side (CPU) checks of the drop-in boundary: module trees, state_dict keys and shapes, sigma tables
and denoiser coefficients, against fixtures captured from the reference."""
import json
from pathlib import Path

import pytest
import torch

import neurosis_amd.modules.diffusion as D
from tests.golden.make_golden import UNET_SD15_TINY, UNET_TINY, VAE_TINY, synth_state_dict
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"


@pytest.mark.parametrize("name,cfg,cls", [("unet_sdxl_tiny", UNET_TINY, "UNetModel"), ("unet_sd15_tiny", UNET_SD15_TINY, "UNetModel"), ("vae_encoder_tiny", VAE_TINY, "Encoder")])
def test_state_dict_keys_and_shapes_match_reference(name, cfg, cls):
    ref = json.loads((G / f"{name}_keys.json").read_text())
    mod = getattr(D, cls)(**cfg)
    mine = {k: list(v.shape) for k, v in mod.state_dict().items()}
    assert list(mine.keys()) == list(ref.keys())
    assert mine == ref
    mod.load_state_dict(synth_state_dict(ref))  # strict


def test_full_size_sdxl_topology_on_meta():
    cfg = dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=True, in_channels=4, out_channels=4, model_channels=320,
               attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64, use_linear_in_transformer=True,
               transformer_depth=[1, 2, 10], context_dim=2048, spatial_transformer_attn_type="softmax-xformers")
    with torch.device("meta"):
        net = D.UNetModel(**cfg)
    n = sum(p.numel() for p in net.parameters())
    assert n == 2_567_463_684, n  # SDXL-base UNet (SURVEY: 2 567.5 M)
    from neurosis_amd.modules.attention import BasicTransformerBlock

    assert sum(isinstance(m, BasicTransformerBlock) for m in net.modules()) == 70


def test_sigma_table_and_denoiser_coefficients():
    fx = load_fixture("unet_sdxl_tiny")
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization())
    assert torch.equal(den.sigmas, fx["sigma_table"])  # 1001 entries, trailing 0.0 (quirk Q1)
    assert not den.sigmas.requires_grad  # detached (quirk Q5)
    c_skip, c_out, c_in, c_noise = den.coefficients(fx["sigma"])
    assert torch.equal(c_noise, fx["c_noise_idx"])
    sq = den.sigmas[c_noise]
    assert torch.allclose(c_out, -sq) and torch.allclose(c_in, 1 / (sq**2 + 1) ** 0.5) and torch.equal(c_skip, torch.ones_like(sq))
    assert torch.allclose(D.EpsWeighting()(fx["sigma"]), fx["sigma"] ** -2.0)


def test_errors_match_reference():
    net = D.UNetModel(in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[], channel_mult=[1], num_heads=1)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 4, 8, 8), torch.zeros(1), None, y=torch.zeros(1, 4))  # y for a non-class-conditional model
    from neurosis_amd.modules.attention import BasicTransformerBlock

    with pytest.raises(ValueError):
        BasicTransformerBlock(32, 2, 16, attn_mode="nope")


def test_product_path_never_imports_the_oracle():
    import ast

    root = Path(__file__).resolve().parent.parent / "neurosis_amd"
    for f in root.rglob("*.py"):
        tree = ast.parse(f.read_text())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.module:
                names = [node.module]
            assert not any(n.split(".")[0] == "oracle" for n in names), f"{f} imports the oracle"
            # ... nor the reference package (LPIPS used to look its weights up through `neurosis.data`)
            assert not any(n.split(".")[0] == "neurosis" for n in names), f"{f} imports the reference package"


def test_two_engines_keep_their_own_state():
    """ADVICE r1 (module-global training flags): the accumulate flag, the zeroed-gradients promise and the side stream belong
    to ONE engine (`ops.EngineState`, reached through the parameter a kernel wrapper is handed), not to the process."""
    from types import SimpleNamespace

    from neurosis_amd import ops

    a, b = ops.EngineState(), ops.EngineState()
    pa, pb, free = torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2)), torch.nn.Parameter(torch.zeros(2))
    pa._nk_store = SimpleNamespace(state=a)
    pb._nk_store = SimpleNamespace(state=b)
    assert ops.state_of(pa) is a and ops.state_of(pb) is b and ops.state_of(free) is ops.state and ops.state_of(None) is ops.state
    a.grad_accumulate = True
    assert ops.wgrad_mode(pa) == 1 and ops.wgrad_mode(pb) == 0 and ops.wgrad_mode(free) == 0
    b.assume_zeroed = True
    assert ops.wgrad_mode(pb) == 2 and ops.wgrad_mode(pa) == 1
    # a channel-padded stand-in parameter: shares its engine's side stream, never accumulates
    a.wgrad_stream = object()
    d = a.derived()
    pad = torch.nn.Parameter(torch.zeros(2))
    pad._nk_state = d
    assert ops.state_of(pad) is d and d.wgrad_stream is a.wgrad_stream and ops.wgrad_mode(pad) == 0
    assert b.wgrad_stream is None and ops.state.wgrad_stream is None


def test_graph_kind_selection_from_the_environment(monkeypatch):
    from neurosis_amd.graphs import graphs_enabled

    monkeypatch.delenv("NK_GRAPH", raising=False)
    assert graphs_enabled("unet") and not graphs_enabled("vae") and not graphs_enabled("te")      # default: the training chain only
    monkeypatch.setenv("NK_GRAPH", "0")
    assert not graphs_enabled("unet")
    monkeypatch.setenv("NK_GRAPH", "1")
    assert graphs_enabled("unet") and graphs_enabled("vae") and graphs_enabled("te")
    monkeypatch.setenv("NK_GRAPH", "unet,te")
    assert graphs_enabled("unet") and graphs_enabled("te") and not graphs_enabled("vae")


def test_sharded_exchange_plan_on_the_full_size_unet():
    """dp.SlicePlan on the real SDXL UNet's parameter list (meta device: shapes only), 8 ranks: every tensor has exactly one owner, every part
    is a run of whole tensors (the split of each slice that minimises its largest part), and the padded staging rows stay moderate -- the
    bytes the reduce-scatter / all-gather move exceed the parameters' own by 14 % in total (the widest tensor, a 10240 x 1280 FeedForward
    projection, is 13 M of a ~40-80 M-element part): 0.75 x 1.14 = 0.86 of the all-reduce's bytes per link."""
    from types import SimpleNamespace

    from neurosis_amd.dp import SlicePlan, _top_block_ranges

    cfg = dict(adm_in_channels=2816, num_classes="sequential", use_checkpoint=False, in_channels=4, out_channels=4, model_channels=320,
               attention_resolutions=[4, 2], num_res_blocks=2, channel_mult=[1, 2, 4], num_head_channels=64, use_linear_in_transformer=True,
               transformer_depth=[1, 2, 10], context_dim=2048, spatial_transformer_attn_type="softmax-xformers")
    with torch.device("meta"):
        net = D.UNetModel(**cfg)
    params = list(net.parameters())
    offsets, total = [], 0
    for p in params:
        offsets.append(total)
        total += (p.numel() + 63) // 64 * 64
    index = {id(p): i for i, p in enumerate(params)}

    def param_range(m):
        idx = [index[id(p)] for p in m.parameters()]
        return (offsets[min(idx)], (offsets + [total])[max(idx) + 1]) if idx else (0, 0)

    store = SimpleNamespace(params=params, offsets=offsets, numel=total, param_range=param_range)
    ranges = sorted(_top_block_ranges(net, store))
    assert ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))      # the blocks tile the buffers
    world = 8
    plans = [SlicePlan(store, lo, hi, world) for lo, hi in ranges]
    owners = [0] * len(params)
    for pl in plans:
        assert pl.cuts[0] == pl.lo and pl.cuts[-1] == pl.hi and sorted(pl.cuts) == pl.cuts
        for r in range(world):
            for t in range(pl.tcuts[r], pl.tcuts[r + 1]):
                owners[t] += 1
    assert all(o == 1 for o in owners)
    staged = sum(world * pl.row for pl in plans)
    per_rank = [sum(pl.sizes[r] for pl in plans) for r in range(world)]
    print(f"rs_ag plan, SDXL UNet, 8 ranks: {len(plans)} slices, staged / owned elements = {staged / total:.4f}, "
          f"largest / smallest share of the optimizer = {max(per_rank) / total:.4f} / {min(per_rank) / total:.4f}")
    assert staged <= 1.16 * total, staged / total
    assert max(per_rank) <= 1.15 * total / world, max(per_rank) * world / total


def test_state_dict_never_communicates_under_the_sharded_exchange():
    """`if rank == 0: torch.save(engine.state_dict())` must not hang in a hidden collective.  With sharded (rs_ag) masters
    that are not whole the call raises and names the fix; sync_masters() is the explicit, all-rank collective."""
    from types import SimpleNamespace

    import pytest

    from neurosis_amd.models.diffusion import DiffusionEngine

    calls = []
    dp = SimpleNamespace(sharded=True, masters_whole=False, sync_masters=lambda: calls.append("sync"))
    eng = SimpleNamespace(join_optimizer=lambda: calls.append("join"), store=SimpleNamespace(dp=dp))
    with pytest.raises(RuntimeError, match="sync_masters"):
        DiffusionEngine.state_dict(eng)
    assert "sync" not in calls
    DiffusionEngine.sync_masters(eng)
    assert calls[-1] == "sync"
