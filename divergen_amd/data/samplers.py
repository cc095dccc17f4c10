"""Rank-sharded index streams for the training / evaluation loaders.

The contract is the reference's (D2/data/samplers/distributed_sampler.py:15-70 TrainingSampler, :129-243
RepeatFactorTrainingSampler, :245-278 InferenceSampler): every rank builds the SAME seeded stream and keeps every
world_size-th element starting at its rank.  What has to match the reference bit for bit is the order of the draws on the one
CPU generator -- `rand(n)` for the fractional repeats, then `randperm(len(epoch))` -- so that a run here visits the images in
the reference's order for the same seed (tests/golden/samplers.npz holds streams of the reference's own classes).
"""
import math

import torch

from ..utils import comm


def _rank_share(stream, rank, world):
    """Every world-th element of an endless stream, starting at `rank`."""
    for k, idx in enumerate(stream):
        if k % world == rank:
            yield idx


class _SeededStream:
    """Seed / rank / world bookkeeping shared by the two training samplers; subclasses provide `epoch(generator)` -> 1-D int64
    tensor of dataset indices for one pass."""

    def __init__(self, shuffle, seed, rank, world_size):
        self.shuffle = bool(shuffle)
        self.seed = int(comm.shared_random_seed() if seed is None else seed)
        self.rank = comm.get_rank() if rank is None else int(rank)
        self.world = comm.get_world_size() if world_size is None else int(world_size)

    def epoch(self, generator):
        raise NotImplementedError

    def stream(self):
        gen = torch.Generator().manual_seed(self.seed)
        while True:
            ids = self.epoch(gen)
            if self.shuffle:
                ids = ids[torch.randperm(ids.numel(), generator=gen)]
            yield from ids.tolist()

    def __iter__(self):
        return _rank_share(self.stream(), self.rank, self.world)


class TrainingSampler(_SeededStream):
    """Endless passes over range(size), each pass a fresh permutation of the shared generator (or 0..size-1 when not shuffling)."""

    def __init__(self, size, shuffle=True, seed=None, rank=None, world_size=None):
        if size <= 0:
            raise ValueError("TrainingSampler needs a non-empty dataset")
        super().__init__(shuffle, seed, rank, world_size)
        self.size = int(size)

    def epoch(self, generator):
        return torch.arange(self.size)


class RepeatFactorTrainingSampler(_SeededStream):
    """LVIS repeat-factor sampling (Gupta et al. 2019, sec. 4.1): image i appears floor(r_i) times per pass plus once more
    with probability frac(r_i)."""

    def __init__(self, repeat_factors, *, shuffle=True, seed=None, rank=None, world_size=None):
        super().__init__(shuffle, seed, rank, world_size)
        rf = torch.as_tensor(repeat_factors, dtype=torch.float32)
        self.whole = torch.trunc(rf)
        self.frac = rf - self.whole

    @staticmethod
    def repeat_factors_from_category_frequency(dataset_dicts, repeat_thresh):
        """r(image) = max over its categories c of max(1, sqrt(t / f(c))), f(c) = share of images that contain c; images
        without annotations get 1."""
        per_image = [{a["category_id"] for a in d["annotations"]} for d in dataset_dicts]
        images_with = {}
        for cats in per_image:
            for cid in cats:
                images_with[cid] = images_with.get(cid, 0) + 1
        n = len(dataset_dicts)
        cat_rep = {cid: max(1.0, math.sqrt(repeat_thresh / (cnt / n))) for cid, cnt in images_with.items()}
        return torch.tensor([max((cat_rep[cid] for cid in cats), default=1.0) for cats in per_image], dtype=torch.float32)

    def epoch(self, generator):
        extra = torch.rand(self.frac.numel(), generator=generator) < self.frac
        counts = (self.whole + extra.to(self.whole.dtype)).to(torch.int64)
        return torch.repeat_interleave(torch.arange(counts.numel()), counts)


class InferenceSampler:
    """Contiguous shards of range(size), sizes differing by at most one (the first size % world ranks take the extra)."""

    def __init__(self, size, rank=None, world_size=None):
        rank = comm.get_rank() if rank is None else int(rank)
        world = comm.get_world_size() if world_size is None else int(world_size)
        self.shard = self._get_local_indices(int(size), world, rank)

    @staticmethod
    def _get_local_indices(total_size, world_size, rank):
        base, extra = divmod(total_size, world_size)
        start = rank * base + min(rank, extra)
        return range(start, start + base + (1 if rank < extra else 0))

    def __iter__(self):
        return iter(self.shard)

    def __len__(self):
        return len(self.shard)
