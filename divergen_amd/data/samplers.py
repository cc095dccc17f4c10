"""Index samplers that shard the training / inference streams across ranks.
Mirrors D2/data/samplers/distributed_sampler.py:15-70 (TrainingSampler), :129-243
(RepeatFactorTrainingSampler, LVIS repeat-factor sampling), :245-278 (InferenceSampler)."""
import itertools
import math
from collections import defaultdict

import torch

from ..utils import comm


class TrainingSampler:
    def __init__(self, size, shuffle=True, seed=None, rank=None, world_size=None):
        assert size > 0
        self._size, self._shuffle = size, shuffle
        self._seed = int(comm.shared_random_seed() if seed is None else seed)
        self._rank = comm.get_rank() if rank is None else rank
        self._world_size = comm.get_world_size() if world_size is None else world_size

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from torch.arange(self._size).tolist()


class RepeatFactorTrainingSampler:
    def __init__(self, repeat_factors, *, shuffle=True, seed=None, rank=None, world_size=None):
        self._shuffle = shuffle
        self._seed = int(comm.shared_random_seed() if seed is None else seed)
        self._rank = comm.get_rank() if rank is None else rank
        self._world_size = comm.get_world_size() if world_size is None else world_size
        self._int_part = torch.trunc(repeat_factors)
        self._frac_part = repeat_factors - self._int_part

    @staticmethod
    def repeat_factors_from_category_frequency(dataset_dicts, repeat_thresh):
        """r(I) = max_{c in I} max(1, sqrt(t / f(c))), f(c) = fraction of images containing c."""
        category_freq = defaultdict(int)
        for d in dataset_dicts:
            for cat_id in {ann["category_id"] for ann in d["annotations"]}:
                category_freq[cat_id] += 1
        n = len(dataset_dicts)
        category_rep = {c: max(1.0, math.sqrt(repeat_thresh / (v / n))) for c, v in category_freq.items()}
        reps = []
        for d in dataset_dicts:
            cats = {ann["category_id"] for ann in d["annotations"]}
            reps.append(max({category_rep[c] for c in cats}, default=1.0))
        return torch.tensor(reps, dtype=torch.float32)

    def _get_epoch_indices(self, generator):
        rands = torch.rand(len(self._frac_part), generator=generator)
        rep = self._int_part + (rands < self._frac_part).float()
        indices = []
        for i, r in enumerate(rep):
            indices.extend([i] * int(r.item()))
        return torch.tensor(indices, dtype=torch.int64)

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            indices = self._get_epoch_indices(g)
            if self._shuffle:
                yield from indices[torch.randperm(len(indices), generator=g)].tolist()
            else:
                yield from indices.tolist()


class InferenceSampler:
    def __init__(self, size, rank=None, world_size=None):
        self._size = size
        self._rank = comm.get_rank() if rank is None else rank
        self._world_size = comm.get_world_size() if world_size is None else world_size
        self._local_indices = self._get_local_indices(size, self._world_size, self._rank)

    @staticmethod
    def _get_local_indices(total_size, world_size, rank):
        shard = total_size // world_size
        left = total_size % world_size
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        return range(begin, min(sum(sizes[:rank + 1]), total_size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)
