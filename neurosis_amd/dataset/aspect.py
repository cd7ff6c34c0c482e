"""Aspect-ratio bucketing: every batch holds images of ONE (width, height) so that it is a dense tensor, and the buckets all
have ~the same pixel count so that every step costs ~the same (the five ~1024^2 SDXL buckets of `bench.py --mixed-res` differ by
< 4 % in tokens).  Behaviour of `neurosis.dataset.aspect.{bucket,lists,sampler}` and of the batch schedule in
`neurosis.dataset.imagefolder.aspect.ImageFolderDataset.get_batch_iterator` (:160-191), without pandas / PIL.

Why this matters on MI355X: a rank's step time is set by its bucket's token count and the flat gradient all-reduce joins all
ranks every step, so the schedule that keeps per-rank work equal is "same pixel budget per bucket", which is what the lists
below encode; nothing in the kernels depends on the bucket beyond (H, W) being multiples of 64 (8 for the VAE x 8 for the UNet).
"""
from __future__ import annotations

import math
from bisect import bisect_left
from dataclasses import dataclass, field
from itertools import product
from typing import Iterator, Optional, Sequence

import numpy as np
import torch


def percent_diff(a: int, b: int) -> float:
    return round((a - b) / ((a + b) / 2) * 100, 2)


@dataclass
class AspectBucket:
    width: int
    height: int
    square_px: Optional[int] = field(default=None, repr=False)
    error: Optional[float] = field(init=False, default=None)

    def __post_init__(self) -> None:
        if self.width % 32 or self.height % 32:
            raise ValueError(f"width and height must be multiples of 32, got {self.width} and {self.height}")
        if self.square_px:
            self.error = percent_diff(self.width * self.height, self.square_px**2)

    aspect = property(lambda self: round(self.width / self.height, 4))
    pixels = property(lambda self: self.width * self.height)
    shape = property(lambda self: (self.height, self.width, 3))      # numpy convention
    size = property(lambda self: (self.width, self.height))          # PIL convention

    def __hash__(self) -> int:
        return hash((self.width, self.height, self.square_px or 0))

    @classmethod
    def flipped(cls, bucket: "AspectBucket") -> "AspectBucket":
        return cls(bucket.height, bucket.width)

    @staticmethod
    def select_by_px(buckets: Sequence["AspectBucket"], alt: bool = False) -> "AspectBucket":
        """the bucket with the most pixels (alt: the runner-up)"""
        if not buckets:
            raise ValueError("Cannot select from empty list of buckets")
        ranked = sorted(buckets, key=lambda b: b.pixels)
        return ranked[-2] if alt and len(ranked) > 1 else ranked[-1]


class AspectBucketList:
    """Buckets generated from edge / aspect / pixel-budget constraints (reference bucket.py:83-239), sorted by aspect."""

    def __init__(self, n_buckets: int = 25, edge_min: int = 512, edge_max: int = 1536, edge_step: int = 64, max_aspect: float = 2.5,
                 tgt_pixels: int = 1024 * 1024, tolerance: float = 5, bias_square: bool = True, use_atan: bool = False):
        if not 1 <= n_buckets <= 100:
            raise ValueError(f"n_buckets must be in [1, 100], got {n_buckets}")
        if edge_min < edge_step or edge_min > edge_max:
            raise ValueError(f"edge_min must be in [edge_step, edge_max], got {edge_min}")
        if edge_max > 4096:
            raise ValueError(f"edge_max must be in [edge_min, 4096], got {edge_max}")
        if edge_max % edge_step or edge_min % edge_step:
            raise ValueError(f"min and max must be multiples of step, got {edge_min} and {edge_max}")
        if edge_max // edge_min < max_aspect:
            raise ValueError(f"max_aspect must be less than edge_max / edge_min, got {max_aspect}")
        self.n_buckets, self.edge_min, self.edge_max, self.edge_step = n_buckets, edge_min, edge_max, edge_step
        self.max_aspect = max_aspect if max_aspect > 0.0 else float("inf")
        self.max_pixels = int(tgt_pixels * (1.0 + tolerance / 100))
        self.min_pixels = int(tgt_pixels * (1.0 - tolerance / 100))
        self.bias_square, self.use_atan = bias_square, use_atan
        side = math.sqrt(tgt_pixels)
        self._square_px = int(side) if side.is_integer() else None
        if not hasattr(self, "data"):          # predefined lists set .data before calling up
            self.data = self._generate()

    def _generate(self) -> list:
        edges = range(self.edge_min, self.edge_max + 1, self.edge_step)
        by_aspect: dict = {}
        for w, h in product(edges, edges):
            if w >= h and self.min_pixels <= w * h <= self.max_pixels and w / h <= self.max_aspect:
                bucket = AspectBucket(w, h, square_px=self._square_px)
                by_aspect.setdefault(round(bucket.aspect, 2), []).append(bucket)
        candidates = sorted((AspectBucket.select_by_px(group) for group in by_aspect.values()), key=lambda b: b.aspect)
        if len(candidates) < self.n_buckets:
            candidates += sorted((AspectBucket.select_by_px(group, alt=True) for group in by_aspect.values()), key=lambda b: b.aspect)
            if len(candidates) < self.n_buckets:
                raise ValueError(f"{self.n_buckets} buckets requested but only {len(candidates)} buckets generated. "
                                 "Try reducing edge_step or edge_min, or increasing edge_max.")
        # landscape picks evenly spread over the candidates, plus their portrait mirrors
        picks = np.linspace(0, len(candidates) - 1, int(np.clip((self.n_buckets + 1) // 2, 1, len(candidates))), dtype=int).tolist()
        chosen = {candidates[i] for i in picks} | {AspectBucket.flipped(candidates[i]) for i in picks}
        return sorted(chosen, key=lambda b: b.aspect)

    # -- container protocol ------------------------------------------------------------------------
    def __len__(self) -> int:
        return len(self.data)

    def __iter__(self):
        return iter(self.data)

    def __getitem__(self, i):
        return self.data[i]

    ratios = property(lambda self: [b.aspect for b in self.data])
    arctans = property(lambda self: [np.arctan(b.aspect) for b in self.data])
    indices = property(lambda self: list(range(len(self.data))))

    # -- lookup ------------------------------------------------------------------------------------
    def bucket_idx(self, ratio: float) -> int:
        """index of the bucket an image of aspect `ratio` (width / height) goes to"""
        if ratio < 0.0:
            raise ValueError(f"ratio must be > 0, got {ratio}")
        if ratio == 1.0:
            return self.ratios.index(1.0)
        key = np.arctan(ratio) if self.use_atan else ratio
        axis = self.arctans if self.use_atan else self.ratios
        if self.bias_square:
            # the neighbour on the square side, so the bucket always fits inside the rescaled image
            return bisect_left(axis, key) - (1 if ratio > 1.0 else 0)
        return int(np.interp(key, axis, self.indices).round().astype(int))

    def bucket(self, ratio: float) -> AspectBucket:
        return self.data[self.bucket_idx(ratio)]


class SDXLBucketList(AspectBucketList):
    """the 40 buckets of the original SDXL training run (reference lists.py:4-69): every 64-px edge pair within 5 % of 1024^2"""

    _TRAIN_RES = 1024
    _PORTRAIT = [(512, 2048), (512, 1984), (512, 1920), (512, 1856), (576, 1792), (576, 1728), (576, 1664), (640, 1600), (640, 1536), (704, 1472),
                 (704, 1408), (704, 1344), (768, 1344), (768, 1280), (832, 1216), (832, 1152), (896, 1152), (896, 1088), (960, 1088), (960, 1024)]

    def __init__(self, bias_square: bool = True, use_atan: bool = False):
        # landscape = the portrait buckets mirrored, except 704x1344, which the original list has in portrait only
        pairs = self._PORTRAIT + [(1024, 1024)] + [(h, w) for w, h in reversed(self._PORTRAIT) if (w, h) != (704, 1344)]
        self.data = [AspectBucket(w, h, self._TRAIN_RES) for w, h in pairs]
        super().__init__(n_buckets=len(self.data), edge_min=512, edge_max=2048, edge_step=64, max_aspect=4.0, tgt_pixels=self._TRAIN_RES**2,
                         tolerance=5, bias_square=bias_square, use_atan=use_atan)


def bucket_batch_schedule(bucket_of_sample: Sequence[int], batch_size: int) -> Iterator[list]:
    """Batches of sample indices, each from one bucket (reference imagefolder/aspect.py:160-191).  Buckets with fewer than one
    batch of samples are skipped; each bucket contributes len // batch_size batches in an order shuffled across buckets; inside a
    bucket, samples are visited in one shared random permutation of positions.  Randomness: numpy's global generator, two
    shuffles, in the reference's order."""
    bucket_of_sample = np.asarray(bucket_of_sample)
    members = {int(b): np.flatnonzero(bucket_of_sample == b) for b in np.unique(bucket_of_sample)}
    visit_order = np.arange(max(len(m) for m in members.values()), dtype=np.int32)
    np.random.shuffle(visit_order)
    members = {b: m for b, m in members.items() if len(m) >= batch_size}
    turns = [b for b, m in members.items() for _ in range(len(m) // batch_size)]
    np.random.shuffle(turns)

    def batches():
        cursor = dict.fromkeys(members, 0)
        for b in turns:
            own, batch = members[b], []
            while len(batch) < batch_size:
                position = visit_order[cursor[b]]
                if position < len(own):
                    batch.append(int(own[position]))
                cursor[b] += 1
            yield batch

    return batches()


class AspectDistributedSampler:
    """Shards the list of BATCHES over the ranks (reference sampler.py:25-87): per epoch a seeded permutation of the batch list,
    padded by wrap-around (or truncated with drop_last) to a multiple of the world size, rank r taking r, r + W, r + 2W, ..."""

    def __init__(self, batches: Sequence[list], num_replicas: Optional[int] = None, rank: Optional[int] = None, shuffle: bool = True, seed: int = 0,
                 drop_last: bool = False):
        if num_replicas is None or rank is None:
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset = list(batches)
        self.num_replicas, self.rank, self.shuffle, self.seed, self.drop_last, self.epoch = num_replicas, rank, shuffle, seed, drop_last, 0
        n = len(self.dataset)
        self.num_samples = n // num_replicas if drop_last and n % num_replicas else math.ceil(n / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.num_samples

    def __iter__(self) -> Iterator[int]:
        n = len(self.dataset)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist() if self.shuffle else list(range(n))
        if self.drop_last:
            order = order[: self.total_size]
        else:
            order = (order * math.ceil(self.total_size / max(n, 1)))[: self.total_size] if self.total_size > n else order
        if len(order) != self.total_size:
            raise ValueError(f"Expected indices to have length {self.total_size}, but got {len(order)}")
        return iter(order[self.rank:self.total_size:self.num_replicas])


def collate_bucket_batch(samples: Sequence[dict], image_key: str = "image", caption_key: str = "caption") -> dict:
    """The batch dictionary DiffusionEngine / GeneralConditioner consume (reference imagefolder/aspect.py:74-100 + collate):
    images stacked to [B, 3, H, W]; captions and the three SDXL size tuples as lists (the conditioner turns the tuples into
    fp32 rows on the device)."""
    first = samples[0]
    batch = {key: [s[key] for s in samples] for key in first}
    shapes = {tuple(s[image_key].shape) for s in samples}
    if len(shapes) != 1:
        raise ValueError(f"a bucketed batch must hold one image shape, got {sorted(shapes)}")
    batch[image_key] = torch.stack([torch.as_tensor(s[image_key]) for s in samples], dim=0)
    return batch
