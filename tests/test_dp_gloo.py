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

    fx = torch.load(G / "unet_sdxl_tiny.pt", weights_only=False)
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
