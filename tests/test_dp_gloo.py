"""The data-parallel exchange (neurosis_amd/dp.py) with world_size 2 on CPU (gloo).

1. FlatGradReducer: slice-wise asynchronous-API all-reduce of a flat gradient buffer gives the element-wise sum,
   whatever the slicing / chunking, including a bf16-on-the-wire variant.
2. Semantics replaced (Lightning DDP): the mean over ranks of per-rank gradients equals the gradient of the
   concatenated batch -- checked with the CPU oracle on a tiny UNet shard pair (loss is a mean of per-sample means,
   so equal shard sizes give the exact mean of means).
"""
import json
import os
import socket
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from tests.golden.fixture_io import load_fixture

G = Path(__file__).resolve().parent / "golden"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_reduce(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from neurosis_amd.dp import FlatGradReducer

    n = 10_007
    base = torch.arange(n, dtype=torch.float32)
    flat = base * (rank + 1)
    red = FlatGradReducer(flat, max_chunk=1000)
    # ranges arrive back to front, as the UNet's backward finalises them
    for lo, hi in [(7000, n), (2500, 7000), (0, 2500)]:
        red.reduce_range(lo, hi)
    red.finish()
    ok = torch.equal(flat, base * sum(r + 1 for r in range(world)))
    flat2 = (base * (rank + 1) / 128).clone()
    red2 = FlatGradReducer(flat2, wire_dtype=torch.bfloat16)
    red2.reduce_range(0, n)
    red2.finish()
    ref = sum((base * (r + 1) / 128).to(torch.bfloat16).float() for r in range(world))
    ok2 = bool((flat2 - ref).abs().max() <= 1e-2 * ref.abs().max())
    out[rank] = (ok, ok2, red.world)
    dist.destroy_process_group()


def test_flat_grad_reducer_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_reduce, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == (True, True, 2), out[r]


def _worker_oracle(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from neurosis_amd.dp import FlatGradReducer
    from oracle import sdxl_oracle as O
    from tests.golden.make_golden import synth_state_dict

    fx = load_fixture("unet_sdxl_tiny")
    shapes = json.loads((G / "unet_sdxl_tiny_keys.json").read_text())
    names = list(shapes)
    table = O.legacy_ddpm_sigmas()

    def grads(sel):
        sd = {k: v.clone().requires_grad_(True) for k, v in synth_state_dict(shapes).items()}
        net = lambda xin, t: O.unet_forward(sd, fx["cfg"], xin, t, fx["context"][sel], fx["y"][sel])
        loss = O.edm_loss(net, table, fx["x"][sel], fx["sigma"][sel], fx["noise"][sel])
        loss.mean().backward()
        return torch.cat([sd[k].grad.reshape(-1) for k in names])

    flat = grads(slice(rank, rank + 1))          # this rank's shard: one sample
    red = FlatGradReducer(flat)
    red.reduce_range(flat.numel() // 2, flat.numel())
    red.reduce_range(0, flat.numel() // 2)
    red.finish()
    flat /= world                                 # the optimizer's grad_scale = 1/world
    full = grads(slice(0, world))                 # single-process gradient of the concatenated batch
    out[rank] = float((flat - full).abs().max() / full.abs().max())
    dist.destroy_process_group()


def test_mean_of_rank_gradients_equals_full_batch_gradient():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_oracle, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] < 1e-5, out[r]


def _worker_sharded(rank, world, port, out):
    """The sharded exchange (mode rs_ag: reduce_scatter_tensor of every slice into tensor-aligned parts, optimizer on the owned parts,
    all_gather_into_tensor of the shadows and of the directly-read fp32 masters) against the all-reduce one, with gloo on CPU: gradient slices
    arrive back to front as the UNet's backward finalises them; each mode then applies the same update rule -- p -= lr * mean gradient --
    (all-reduce: everywhere; rs_ag: the owner on its parts; masters gathered on demand).  Masters and shadows must agree exactly."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace

    from neurosis_amd.dp import FlatDataParallel, SlicePlan

    torch.manual_seed(0)
    sizes = [1000, 37, 4096, 512, 2048, 64, 3000, 800, 640, 640, 640]
    offsets, total = [], 0
    for n in sizes:
        offsets.append(total)
        total += (n + 63) // 64 * 64
    ends = offsets + [total]
    blocks = [(offsets[8], total), (offsets[5], offsets[8]), (offsets[2], offsets[5]), (0, offsets[2])]       # four "blocks", back to front
    vectors = (1, 5)

    def make_store():
        master = torch.linspace(-1, 1, total)
        store = SimpleNamespace(params=[torch.empty(n) if i in vectors else torch.empty(n // 8, 8) for i, n in enumerate(sizes)], offsets=offsets, numel=total,
                                master=master.clone(), shadow=master.to(torch.bfloat16), grad=torch.zeros(total), listeners=[],
                                state=SimpleNamespace(wgrad_stream=None, aux_stream=None))
        store._mark_fresh = lambda: None
        store.refresh = lambda: store.shadow.copy_(store.master.to(torch.bfloat16))
        return store

    # the plan of one slice: tensor-aligned cuts, every element owned exactly once, parts of about equal size
    pl = SlicePlan(make_store(), offsets[2], offsets[5], world)
    ok_plan = (pl.cuts[0] == offsets[2] and pl.cuts[-1] == offsets[5] and all(c in ends for c in pl.cuts) and sorted(pl.cuts) == pl.cuts
               and all(pl.cuts[r] == ends[pl.tcuts[r]] for r in range(world + 1)) and pl.row % 64 == 0 and pl.row >= max(pl.sizes))
    # the last block is three equal tensors: at world 3 its parts are equal and aligned -> no staging copies (the slice is its own staging layout)
    pl_eq = SlicePlan(make_store(), offsets[8], total, world)
    ok_plan = ok_plan and (world != 3 or all(sz == pl_eq.row for sz in pl_eq.sizes))
    # world 4: every block of this toy set has FEWER tensors than ranks (3, 3, 3, 2) -- the last ranks own nothing of a slice (SDXL's
    # time_embed / label_emb / input_blocks.0 / out at 8 ranks), yet take part in its reduce-scatter and all-gather with an empty part
    if world > 3:
        ok_plan = ok_plan and pl.sizes[-1] == 0 and sum(pl.sizes) == offsets[5] - offsets[2] and pl.tcuts[-1] == pl.tcuts[-2]
    results = {}
    for mode in ("allreduce", "rs_ag"):
        store = make_store()
        unet = SimpleNamespace(grad_ready_hook=None)
        dp = FlatDataParallel(unet, store, mode=mode, broadcast_params=False, slices=blocks)
        # a stand-in for the fused optimizer's statistics: one fp32 value per element, laid out in tensor order like FlatAdafactor.state
        opt = SimpleNamespace(state=torch.zeros(total), restrict_ranges=lambda r: None, state_span=lambda lo, hi: (ends[lo], ends[hi]))
        dp.attach_optimizer(opt)
        owned = dp.owned_ranges()
        g = torch.Generator().manual_seed(100 + rank)
        for step in range(3):
            store.grad.copy_(torch.randn(total, generator=g))
            for lo, hi in blocks:
                dp.exchange_range(lo, hi)
            scale = dp.finish()
            for lo_t, hi_t in owned:                                                                # the "optimizer", on the owned parts
                a, b = ends[lo_t], ends[hi_t]
                store.master[a:b] -= 0.1 * scale * store.grad[a:b]
                store.shadow[a:b] = store.master[a:b].to(torch.bfloat16)
                opt.state[a:b] += (scale * store.grad[a:b]) ** 2
            dp.after_optimizer_step()
        shadows_before_sync = store.shadow.clone()
        # the 1-D parameters, which kernels read as fp32 masters, must be current on every rank BEFORE any sync_masters()
        vec_before_sync = torch.cat([store.master[offsets[t]:offsets[t] + sizes[t]] for t in vectors]).clone()
        # masters_whole: what DiffusionEngine.state_dict() consults instead of communicating (a rank-local state_dict() must not deadlock)
        whole_flags = [dp.masters_whole]
        dp.sync_masters()
        whole_flags.append(dp.masters_whole)
        ok_plan = ok_plan and whole_flags == ([False, True] if mode == "rs_ag" else [True, True])
        results[mode] = (store.master.clone(), shadows_before_sync, dp.sharded, dp.reducer.take_counts(), opt.state.clone(), vec_before_sync, owned)
    # (two ranks: a + b has one order -> bit-equal; three: the ring all-reduce and the reduce-scatter add in different orders -> fp32 rounding)
    same = torch.equal if world == 2 else (lambda a, b: torch.allclose(a, b, rtol=1e-5, atol=1e-6))
    same_master = same(results["allreduce"][0], results["rs_ag"][0]) and same(results["allreduce"][4], results["rs_ag"][4])
    same_shadow = (same(results["allreduce"][1].float(), results["rs_ag"][1].float()) if world == 2 else
                   torch.allclose(results["allreduce"][1].float(), results["rs_ag"][1].float(), rtol=1e-2, atol=1e-3)) and same(results["allreduce"][5], results["rs_ag"][5])
    # every tensor has exactly one owner across the ranks
    gathered = [None] * world
    dist.all_gather_object(gathered, results["rs_ag"][6])
    owners = [sum(any(a <= t < b for a, b in rr) for rr in gathered) for t in range(len(sizes))]
    one_owner = all(o == 1 for o in owners)
    # (bytes per rank: the padded staging rows of this toy set -- tensors of very unequal size -- hide the 6/8; checked on equal parts below
    # and on the real UNet's parameter list in test_host_logic.py)
    out[rank] = (ok_plan, same_master, same_shadow, results["rs_ag"][2], not results["allreduce"][2], one_owner)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_exchange_equals_allreduce(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sharded, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        assert out[r] == (True,) * 6, out[r]


def _worker_bytes(rank, world, port, out):
    """equal, aligned parts: rs_ag moves exactly 6/8 of the all-reduce's bytes per rank (fp32 reduce-scatter + bf16 all-gather vs fp32 both ways)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace

    from neurosis_amd.dp import FlatDataParallel

    n, k = 1024, 2 * world
    offsets, total = [i * n for i in range(k)], k * n
    counts = {}
    for mode in ("allreduce", "rs_ag"):
        store = SimpleNamespace(params=[torch.empty(n // 8, 8) for _ in range(k)], offsets=offsets, numel=total, master=torch.zeros(total), shadow=torch.zeros(total, dtype=torch.bfloat16),
                                grad=torch.ones(total), listeners=[], state=SimpleNamespace(wgrad_stream=None, aux_stream=None))
        store._mark_fresh = lambda: None
        dp = FlatDataParallel(SimpleNamespace(grad_ready_hook=None), store, mode=mode, broadcast_params=False, slices=[(0, total)])
        dp.exchange_range(0, total)
        dp.finish()
        dp.after_optimizer_step()
        counts[mode] = dp.reducer.take_counts()
        ok_sum = bool((store.grad[dp.plans[(0, total)].cuts[rank]:dp.plans[(0, total)].cuts[rank + 1]] == world).all()) if dp.sharded else bool((store.grad == world).all())
        counts[mode + "_ok"] = ok_sum
    out[rank] = (counts["rs_ag"][1] / counts["allreduce"][1], counts["rs_ag"][0], counts["allreduce_ok"], counts["rs_ag_ok"])
    dist.destroy_process_group()


def test_sharded_exchange_moves_three_quarters_of_the_bytes():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_bytes, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ratio, ncoll, ok_a, ok_s = out[r]
        assert abs(ratio - 0.75) < 1e-6 and ncoll == 2 and ok_a and ok_s, out[r]      # (no 1-D parameters here: one reduce-scatter, one all-gather)
