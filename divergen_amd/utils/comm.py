"""Rank / world helpers and object collectives.  Surface of D2/utils/comm.py:19-199.
One process per GPU; tensors go over RCCL (backend "nccl" on ROCm), pickled objects over gloo."""
import functools

import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_local_rank():
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    if _LOCAL_PROCESS_GROUP is None:
        import os
        return int(os.environ.get("LOCAL_RANK", 0))
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP)


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


@functools.lru_cache()
def _get_global_gloo_group():
    """comm.py:87-96: side group for pickled-object collectives."""
    if dist.get_backend() == "nccl":
        return dist.new_group(backend="gloo")
    return dist.group.WORLD


def all_gather(data, group=None):
    if get_world_size() == 1:
        return [data]
    group = group or _get_global_gloo_group()
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, data, group=group)
    return out


def gather(data, dst=0, group=None):
    if get_world_size() == 1:
        return [data]
    group = group or _get_global_gloo_group()
    if dist.get_rank(group) == dst:
        out = [None] * dist.get_world_size(group)
        dist.gather_object(data, out, dst=dst, group=group)
        return out
    dist.gather_object(data, None, dst=dst, group=group)
    return []


def shared_random_seed():
    """comm.py:156-167: all ranks agree on rank 0's random seed."""
    import numpy as np
    ints = np.random.randint(2 ** 31)
    return all_gather(ints)[0]


def reduce_dict(input_dict, average=True):
    """comm.py:170-199: reduce a dict of scalar tensors to rank 0."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        dist.reduce(values, dst=0)
        if dist.get_rank() == 0 and average:
            values /= world
        return {k: v for k, v in zip(names, values)}
