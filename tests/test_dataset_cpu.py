"""Data side of the mixed-resolution configuration: bucket lists, ratio -> bucket lookup, the bucketed batch
schedule and the distributed sampler against values captured from the reference's classes (tests/golden/make_golden.py::
dataset_cases).  Integer / index work: exact equality."""
from pathlib import Path

import numpy as np
import pytest
import torch

from neurosis_amd.dataset import AspectBucketList, AspectDistributedSampler, SDXLBucketList, bucket_batch_schedule, collate_bucket_batch
from tests.golden.make_golden import BUCKET_LIST_CASES, BUCKET_RATIOS, bucket_assignment
from tests.golden.fixture_io import load_fixture

G = load_fixture("dataset_aspect")


def _build(name):
    if name.startswith("sdxl"):
        return SDXLBucketList(**{"sdxl": {}, "sdxl_atan": dict(use_atan=True), "sdxl_interp": dict(bias_square=False)}[name])
    return AspectBucketList(**BUCKET_LIST_CASES[name])


@pytest.mark.parametrize("name", sorted(G["lists"]))
def test_bucket_lists_and_lookup(name):
    want = G["lists"][name]
    if isinstance(want, str):          # the reference raises for this configuration (its own defaults do)
        with pytest.raises(ValueError) as err:
            _build(name)
        assert f"ValueError: {err.value}" == want
        return
    lst = _build(name)
    assert [(b.width, b.height, b.error) for b in lst] == want
    assert [int(lst.bucket_idx(r)) for r in BUCKET_RATIOS] == G["lookup"][name]
    assert all(b.width % 64 == 0 and b.height % 64 == 0 for b in lst)      # what the VAE (x8) + UNet (x8) need


@pytest.mark.parametrize("batch_size", [4, 16])
def test_bucket_batch_schedule(batch_size):
    np.random.seed(1234)
    got = list(bucket_batch_schedule(bucket_assignment(), batch_size))
    assert got == G["schedule"][batch_size]
    assignment = bucket_assignment()
    for batch in got:                                       # one bucket per batch, no sample twice
        assert len({int(assignment[i]) for i in batch}) == 1
    flat = [i for b in got for i in b]
    assert len(flat) == len(set(flat))


@pytest.mark.parametrize("key", sorted(G["sampler"]))
def test_distributed_sampler(key):
    world, drop_last, shuffle = key
    batches = G["schedule"][4]
    for rank in range(world):
        sampler = AspectDistributedSampler(batches, num_replicas=world, rank=rank, shuffle=shuffle, seed=11, drop_last=drop_last)
        for epoch, want in zip((0, 3), G["sampler"][key][rank]):
            sampler.set_epoch(epoch)
            assert list(sampler) == want and len(sampler) == len(want)
    # the ranks together cover every batch (drop_last: all but a remainder) and each takes the same number
    sampler_sets = [set(G["sampler"][key][r][0]) for r in range(world)]
    assert len(set().union(*sampler_sets)) >= len(batches) - (world - 1 if drop_last else 0)


def test_collate_and_engine_schema():
    buckets = SDXLBucketList()
    assert buckets.bucket(832 / 1216).size == (832, 1152)     # bias_square: the neighbour on the square side of 0.684
    bucket = buckets[14]
    assert bucket.size == (832, 1216)
    samples = [{"image": torch.zeros(3, bucket.height, bucket.width), "caption": f"c{i}", "original_size_as_tuple": (900, 1300),
                "crop_coords_top_left": (0, 4), "target_size_as_tuple": bucket.size} for i in range(4)]
    batch = collate_bucket_batch(samples)
    assert batch["image"].shape == (4, 3, 1216, 832) and batch["caption"] == ["c0", "c1", "c2", "c3"]
    assert batch["target_size_as_tuple"] == [(832, 1216)] * 4
    samples[1]["image"] = torch.zeros(3, 64, 64)
    with pytest.raises(ValueError):
        collate_bucket_batch(samples)
