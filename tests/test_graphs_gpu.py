"""hipGraph replay of the UNet chain (neurosis_amd/graphs.py) against the eager chain it was captured from: same kernels, same
order, so the per-sample loss must be bit-identical and the gradients equal up to the fp32 atomics of the few split-K
weight gradients ("no change in loss bits")."""
import json
import os
from pathlib import Path

import pytest
import torch

from tests.golden.make_golden import UNET_TINY, synth_state_dict

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


def _setup(**unet_kw):
    import neurosis_amd.modules.diffusion as D
    from neurosis_amd.nn import FlatParamStore

    net = D.UNetModel(**{**UNET_TINY, **unet_kw})
    net.load_state_dict(synth_state_dict(json.loads((G / "unet_sdxl_tiny_keys.json").read_text())))
    net = net.cuda()
    store = FlatParamStore(net.parameters())
    store.state.wgrad_stream = torch.cuda.Stream()
    den = D.DiscreteDenoiser(preconditioning=D.EpsPreconditioning(), num_idx=1000, discretization=D.LegacyDDPMDiscretization()).cuda()
    lossfn = D.StandardDiffusionLoss(sigma_generator=D.InjectedSigmaGenerator(), loss_weighting=D.EpsWeighting())
    return net, store, den, lossfn, D.OpenAIWrapper(net)


def _batches(n, hw=(16, 16)):
    g = torch.Generator().manual_seed(7)
    out = []
    for _ in range(n):
        out.append(dict(x=torch.randn(2, 4, *hw, generator=g).cuda(), noise=torch.randn(2, 4, *hw, generator=g).cuda(),
                        sigma=(torch.rand(2, generator=g) * 5 + 0.1).cuda(), ctx=torch.randn(2, 77, UNET_TINY["context_dim"], generator=g).cuda(),
                        y=torch.randn(2, UNET_TINY["adm_in_channels"], generator=g).cuda()))
    return out


def _steps(batches, graph: bool, **unet_kw):
    os.environ["NK_GRAPH"] = "1" if graph else "0"
    try:
        net, store, den, lossfn, wrapped = _setup(**unet_kw)
        losses, grads = [], []
        for b in batches:
            loss = lossfn._forward(wrapped, den, {"crossattn": b["ctx"], "vector": b["y"]}, b["x"], {}, sigmas=b["sigma"], noise=b["noise"])
            loss.mean().backward()
            torch.cuda.synchronize()
            losses.append(loss.detach().clone())
            grads.append(store.grad.clone())
            store.adamw_step(1e-3, (0.9, 0.999), 1e-8, 0.0, 1.0)     # the weights MOVE between steps: a replay must see the new ones
        return net, losses, grads
    finally:
        os.environ.pop("NK_GRAPH", None)


def test_graph_replay_reproduces_the_eager_chain_bit_for_bit():
    batches = _batches(5)
    net_e, loss_e, grad_e = _steps(batches, graph=False)
    net_g, loss_g, grad_g = _steps(batches, graph=True)
    assert net_e._nk_graphs is None
    cg = net_g._nk_graphs
    assert cg is not None and len(cg.pairs) == 1
    pair = next(iter(cg.pairs.values()))
    assert pair.g_f is not None and pair.segments is not None and pair.closure is None
    # one M graph per top-level block (+ head and tail); the blocks that own weights also have a side-stream graph W
    assert len(pair.segments) >= len(net_g.input_blocks) + len(net_g.output_blocks) + 2 and sum(w is not None for _, w, _ in pair.segments) >= len(net_g.input_blocks)
    assert int(cg.ticks) == 4 and cg.replays == 4 * (1 + len(pair.segments))     # warm-up step eager; capture step and three more replayed
    for i, (a, b) in enumerate(zip(loss_e, loss_g)):
        assert torch.equal(a, b), (i, a.tolist(), b.tolist())
    for i, (a, b) in enumerate(zip(grad_e, grad_g)):
        assert float((a - b).norm() / a.norm()) <= 1e-5, i
        assert float(a.norm()) > 0
    # ... and both follow the weights: a network built COLD from the trained state_dict (no cache of any weight-derived tensor:
    # bf16 shadows, channel-padded stand-ins of the 4-channel convolutions) gives the replay's loss on the next batch, bit for bit
    nxt = _batches(6)[-1]
    os.environ["NK_GRAPH"] = "0"
    try:
        cold, cstore, den, lossfn, cwrapped = _setup()
        cold.load_state_dict(net_g.state_dict())
        cstore.refresh()
        want = lossfn._forward(cwrapped, den, {"crossattn": nxt["ctx"], "vector": nxt["y"]}, nxt["x"], {}, sigmas=nxt["sigma"], noise=nxt["noise"]).detach()
        os.environ["NK_GRAPH"] = "1"
        import neurosis_amd.modules.diffusion as D
        got = lossfn._forward(D.OpenAIWrapper(net_g), den, {"crossattn": nxt["ctx"], "vector": nxt["y"]}, nxt["x"], {}, sigmas=nxt["sigma"], noise=nxt["noise"]).detach()
        os.environ["NK_GRAPH"] = "0"
        got_e = lossfn._forward(D.OpenAIWrapper(net_e), den, {"crossattn": nxt["ctx"], "vector": nxt["y"]}, nxt["x"], {}, sigmas=nxt["sigma"], noise=nxt["noise"]).detach()
    finally:
        os.environ.pop("NK_GRAPH", None)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_e, want), (got.tolist(), got_e.tolist(), want.tolist())


@pytest.mark.parametrize("sk", [None, "2"])
def test_tail_weight_gradient_graphs_may_replay_on_the_main_stream(sk, monkeypatch):
    """ChainGraphs._replay_backward shares the W graphs that start behind the end of the main chain out between both streams (measured once,
    by events).  Here the plan is FORCED to every W graph of the second half of the backward: same kernels, same arguments, so the gradients
    must equal the eager chain's, and the W graphs have pools of their own (two of them run at once).
    sk = "2" (ADVICE round 4): NK_GEMM_SK=2 sends fp32 weight gradients to the stream-K kernel, whose workspace is keyed by the capture
    stream -- two W graphs at once would share it, so the replay must IGNORE the plan under that switch; the gradients stay right."""
    if sk is not None:
        monkeypatch.setenv("NK_GEMM_SK", sk)
    batches = _batches(6)
    _, loss_e, grad_e = _steps(batches, graph=False)
    os.environ["NK_GRAPH"] = "1"
    try:
        net, store, den, lossfn, wrapped = _setup()
        grads = []
        for i, b in enumerate(batches):
            loss = lossfn._forward(wrapped, den, {"crossattn": b["ctx"], "vector": b["y"]}, b["x"], {}, sigmas=b["sigma"], noise=b["noise"])
            loss.mean().backward()
            torch.cuda.synchronize()
            assert torch.equal(loss.detach(), loss_e[i])
            grads.append(store.grad.clone())
            store.adamw_step(1e-3, (0.9, 0.999), 1e-8, 0.0, 1.0)
            cg = net._nk_graphs
            pair = next(iter(cg.pairs.values())) if cg is not None and cg.pairs else None
            if pair is not None and pair.segments is not None and i == 2:
                n = len(pair.segments)
                pair.tail = frozenset(k for k, (_, w, _) in enumerate(pair.segments) if w is not None and k >= n // 2)
                pair.cal = None
                assert len(pair.tail) >= 3
    finally:
        os.environ.pop("NK_GRAPH", None)
    assert pair.tail is not None and len(pair.tail) >= 3
    for i, (a, b) in enumerate(zip(grad_e, grads)):
        assert float((a - b).norm() / a.norm()) <= 1e-5, i


def test_each_input_signature_gets_its_own_pair_and_they_share_one_pool():
    """Aspect buckets: two resolutions alternate; each is captured on its second appearance and replayed afterwards."""
    sizes = [(16, 16), (8, 24), (16, 16), (8, 24), (16, 16), (8, 24)]
    batches = [_batches(1, hw)[0] for hw in sizes]
    net_e, loss_e, grad_e = _steps(batches, graph=False)
    net_g, loss_g, grad_g = _steps(batches, graph=True)
    cg = net_g._nk_graphs
    assert len(cg.pairs) == 2 and all(p.segments is not None for p in cg.pairs.values()) and int(cg.ticks) == 4
    for a, b in zip(loss_e, loss_g):
        assert torch.equal(a, b)
    for a, b in zip(grad_e, grad_g):
        assert float((a - b).norm() / a.norm()) <= 1e-5


def test_the_gradient_ready_hook_fires_between_replayed_segments_as_in_the_eager_chain():
    """The data-parallel exchange hangs on UNetModel.grad_ready_hook: the replay must call it once per top-level block, in
    backward order, like the eager chain (and not at all while capturing)."""
    batches = _batches(4)
    seen = {}
    for graph in (False, True):
        os.environ["NK_GRAPH"] = "1" if graph else "0"
        try:
            net, store, den, lossfn, wrapped = _setup()
            per_step = []
            net.grad_ready_hook = lambda m: per_step[-1].append(id(m))
            order = {id(m): i for i, m in enumerate(net.modules())}
            for b in batches:
                per_step.append([])
                lossfn._forward(wrapped, den, {"crossattn": b["ctx"], "vector": b["y"]}, b["x"], {}, sigmas=b["sigma"], noise=b["noise"]).mean().backward()
            torch.cuda.synchronize()
            seen[graph] = [[order[i] for i in s] for s in per_step]
            assert (net._nk_graphs is not None) is graph
        finally:
            os.environ.pop("NK_GRAPH", None)
    assert seen[True] == seen[False] and len(seen[True][0]) >= 6 and all(s == seen[True][0] for s in seen[True])


def test_accumulate_mode_is_part_of_the_signature():
    """The overwrite / add mode of the weight-gradient kernels is a launch argument: a graph captured in one mode must not be
    replayed in the other."""
    batches = _batches(6)
    os.environ["NK_GRAPH"] = "1"
    try:
        net, store, den, lossfn, wrapped = _setup()
        ref_net, ref_store, *_ = _setup()

        def run(n, st, w, b, acc):
            st.state.grad_accumulate = acc
            lossfn._forward(w, den, {"crossattn": b["ctx"], "vector": b["y"]}, b["x"], {}, sigmas=b["sigma"], noise=b["noise"]).mean().backward()
            torch.cuda.synchronize()

        import neurosis_amd.modules.diffusion as D
        # three micro-batches per optimizer step (overwrite, add, add): the "add" signature is captured INSIDE the first step,
        # while the channel-padded stand-ins of the 4-channel convolutions are still valid for that step's weights -- every
        # captured forward must refill them all the same, or the replays after the next optimizer step read stale (freed) ones
        batches = _batches(12)
        for i, b in enumerate(batches):
            run(net, store, wrapped, b, acc=bool(i % 3))
            if i % 3 == 2:
                store.adamw_step(1e-3, (0.9, 0.999), 1e-8, 0.0, 1.0)
        os.environ["NK_GRAPH"] = "0"
        for i, b in enumerate(batches):
            run(ref_net, ref_store, D.OpenAIWrapper(ref_net), b, acc=bool(i % 3))
            if i % 3 == 2:
                ref_store.adamw_step(1e-3, (0.9, 0.999), 1e-8, 0.0, 1.0)
        assert len(net._nk_graphs.pairs) == 2 and all(p.segments is not None for p in net._nk_graphs.pairs.values())
        assert torch.isfinite(store.grad).all()
        assert float((store.grad - ref_store.grad).norm() / ref_store.grad.norm()) <= 1e-4
        assert float((store.master - ref_store.master).norm() / ref_store.master.norm()) <= 1e-5
    finally:
        os.environ.pop("NK_GRAPH", None)


def test_mode_change_or_another_forward_between_forward_and_backward_is_refused():
    """(advisor, round 2) A replayed backward has its overwrite / add mode baked in, and every signature of a chain shares one activation
    pool: flipping the mode after the forward, or running another forward (of any signature) before the backward, must raise -- not
    silently replay the wrong mode or read overwritten activations."""
    batches = _batches(5)
    os.environ["NK_GRAPH"] = "1"
    try:
        net, store, den, lossfn, wrapped = _setup()

        def fwd(b):
            return lossfn._forward(wrapped, den, {"crossattn": b["ctx"], "vector": b["y"]}, b["x"], {}, sigmas=b["sigma"], noise=b["noise"]).mean()

        for b in batches[:3]:              # eager, capture, replay
            store.state.grad_accumulate = False
            fwd(b).backward()
        store.state.grad_accumulate = False
        loss = fwd(batches[3])
        store.state.grad_accumulate = True
        with pytest.raises(RuntimeError, match="accumulate mode changed"):
            loss.backward()
        store.state.grad_accumulate = False
        loss = fwd(batches[3])
        with torch.no_grad():              # another signature (no-grad forward: an evaluation pass) of the same chain, same pool
            fwd(batches[4]); fwd(batches[4]); fwd(batches[4])
        with pytest.raises(RuntimeError, match="has since been replayed"):
            loss.backward()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("NK_GRAPH", None)


def test_split_k_weight_gradient_with_a_multi_megabyte_destination_survives_replay():
    """A 640 x 640 weight gradient over 16 384 rows is a 25-tile grid: split-K with fp32 atomics into a zeroed 1.6 MB destination.
    With `hipMemsetAsync` as the zero-fill, the captured memset node was not ordered before the GEMM on this ROCm and replayed
    gradients were garbage (only beyond ~1 MB: the tiny UNet never showed it); the fill is a kernel now."""
    from neurosis_amd import ops
    from neurosis_amd.graphs import ChainGraphs
    from neurosis_amd.nn import FlatParamStore

    torch.manual_seed(0)
    lin = torch.nn.Linear(640, 640).cuda()
    store = FlatParamStore(lin.parameters())
    store.state.wgrad_stream = torch.cuda.Stream()
    cg = ChainGraphs(lin.weight)

    def fwd(x):
        y, b = ops.linear_fwd(x, lin.weight, lin.bias)

        def bwd(dy):
            est = ops.state_of(lin.weight)
            dx = b(dy)
            if est.segment_hook is not None:
                est.segment_hook(lin)
            ops.join_wgrad_stream(lin.weight)
            return dx

        return y, bwd

    g = torch.Generator().manual_seed(1)
    for step in range(4):
        x = torch.randn(16384, 640, generator=g).cuda().bfloat16()
        dy = torch.randn(16384, 640, generator=g).cuda().bfloat16()
        store.grad.fill_(float("nan"))
        y, bwd = cg.run(fwd, [x])
        dx = bwd(dy)
        torch.cuda.synchronize()
        want = dy.float().t() @ x.float()
        got = lin.weight.grad
        assert torch.isfinite(got).all(), step
        assert float((got - want).norm() / want.norm()) <= 1e-2, step
        assert float((lin.bias.grad - dy.float().sum(0)).norm() / dy.float().sum(0).norm()) <= 1e-2, step
        assert float((y.float() - (x.float() @ lin.weight.detach().bfloat16().float().t() + lin.bias.detach())).abs().max()) <= 0.25
        assert float((dx.float() - dy.float() @ lin.weight.detach().bfloat16().float()).abs().max()) <= 0.5
    assert next(iter(cg.pairs.values())).segments is not None and int(cg.ticks) == 3


def test_frozen_encoder_and_text_towers_replay_bit_for_bit_and_follow_a_weight_reload():
    """ForwardGraphs: the frozen VAE encoder and both text towers.  Outputs handed out are clones (a later replay must not change
    them), and reloading weights invalidates the captured graphs (they read cached bf16 shadows of the old tensors)."""
    from neurosis_amd.models.text_encoder.clip import CLIPTextTower, OpenCLIPTextTower
    from neurosis_amd.modules.diffusion.model import Encoder
    from tests.golden.make_golden import VAE_TINY

    torch.manual_seed(0)
    enc = Encoder(**{k: v for k, v in VAE_TINY.items() if k != "embed_dim"}).cuda().requires_grad_(False)
    hf = CLIPTextTower(vocab_size=1000, hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2).cuda().requires_grad_(False)
    oc = OpenCLIPTextTower(vocab_size=1000, width=64, layers=2, heads=2, embed_dim=64).cuda().requires_grad_(False)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.rand(2, 3, 32, 32, generator=g).cuda() * 2 - 1 for _ in range(4)]
    ids = [torch.randint(1, 999, (2, 77), generator=g).cuda() for _ in range(4)]

    def run_all(graph):
        os.environ["NK_GRAPH"] = "1" if graph else "0"
        try:
            outs = []
            for im, tk in zip(imgs, ids):
                a = enc(im, regularize=True)
                b = hf(tk, output_hidden_states=True)
                c = oc(tk)
                outs.append((a, b["last_hidden_state"], b["pooler_output"], b["hidden_states"][1], c["penultimate"], c["pooled"]))
            torch.cuda.synchronize()
            return outs
        finally:
            os.environ.pop("NK_GRAPH", None)

    eager = run_all(False)
    graphed = run_all(True)
    for m in (enc, hf, oc):
        assert m.__dict__["_nk_fgraphs"].replays == 3          # warm-up eager, then capture + two replays
    for i, (e, gq) in enumerate(zip(eager, graphed)):
        for a, b in zip(e, gq):
            assert torch.equal(a, b), i
    # new weights -> new stamp -> a fresh warm-up / capture, and the new weights' outputs
    with torch.no_grad():
        for p in hf.parameters():
            p.mul_(1.5)
    os.environ["NK_GRAPH"] = "1"
    try:
        after = [hf(ids[0])["last_hidden_state"] for _ in range(3)]
    finally:
        os.environ.pop("NK_GRAPH", None)
    os.environ["NK_GRAPH"] = "0"
    try:
        want = hf(ids[0])["last_hidden_state"]
    finally:
        os.environ.pop("NK_GRAPH", None)
    assert all(torch.equal(a, want) for a in after) and not torch.equal(want, eager[0][1])


def test_a_failed_capture_leaves_the_chain_on_the_eager_launch_path(monkeypatch):
    """A runtime that cannot capture (here: capture_begin made to fail) must not take the training step down: the chain warns
    once and keeps launching its HIP kernels from Python; results unchanged."""
    batches = _batches(4)
    net_e, loss_e, grad_e = _steps(batches, graph=False)

    def boom(self, *a, **k):
        raise RuntimeError("simulated: stream capture unsupported")

    monkeypatch.setattr(torch.cuda.CUDAGraph, "capture_begin", boom)
    with pytest.warns(UserWarning, match="hipGraph capture of the forward chain failed"):
        net_g, loss_g, grad_g = _steps(batches, graph=True)
    assert net_g._nk_graphs.broken and net_g._nk_graphs.replays == 0
    for a, b in zip(loss_e, loss_g):
        assert torch.equal(a, b)
    for a, b in zip(grad_e, grad_g):
        assert float((a - b).norm() / a.norm()) <= 1e-5


def test_activation_checkpointing_replays_too():
    """use_checkpoint=True (the reference's flag): blocks drop their activations and re-run their forward inside backward -- under a
    capture those recomputations are simply more launches of the backward segments."""
    batches = _batches(4)
    net_e, loss_e, grad_e = _steps(batches, graph=False, use_checkpoint=True)
    net_g, loss_g, grad_g = _steps(batches, graph=True, use_checkpoint=True)
    assert next(iter(net_g._nk_graphs.pairs.values())).segments is not None
    for a, b in zip(loss_e, loss_g):
        assert torch.equal(a, b)
    for a, b in zip(grad_e, grad_g):
        assert float((a - b).norm() / a.norm()) <= 1e-5
