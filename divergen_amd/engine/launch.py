"""Process bring-up.  Mirrors D2/engine/launch.py:27-126 and D2/engine/defaults.py:82-144.
One process per GPU; torch.distributed backend "nccl" (= RCCL on ROCm) for GPU runs, "gloo" for CPU
tests.  --num-gpus 0 means auto-detect (DiverGen's launch.sh counts GPUs with nvidia-smi, which
prints nothing on a ROCm box, so the unchanged script passes 0)."""
import argparse
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def default_argument_parser(epilog=None):
    p = argparse.ArgumentParser(epilog=epilog, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    p.add_argument("--resume", action="store_true", help="resume from the checkpoint directory")
    p.add_argument("--eval-only", action="store_true", help="perform evaluation only")
    p.add_argument("--num-gpus", type=int, default=1, help="number of gpus *per machine* (0 = all visible)")
    p.add_argument("--num-machines", type=int, default=1)
    p.add_argument("--machine-rank", type=int, default=0)
    port = 2 ** 15 + 2 ** 14 + hash(os.getuid() if sys.platform != "win32" else 1) % 2 ** 14
    p.add_argument("--dist-url", default="tcp://127.0.0.1:{}".format(port))
    p.add_argument("opts", default=None, nargs=argparse.REMAINDER,
                   help="Modify config options at the end of the command, 'KEY VALUE' pairs")
    return p


def _worker(local_rank, main_func, world_size, gpus_per_machine, machine_rank, dist_url, backend, args):
    rank = machine_rank * gpus_per_machine + local_rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")          # CUs the gradient all-reduce may take from the overlapped backward
    dist.init_process_group(backend=backend, init_method=dist_url, world_size=world_size, rank=rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.barrier()
    try:
        main_func(*args)
    finally:
        dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=(), backend=None):
    if num_gpus_per_machine == 0:
        num_gpus_per_machine = max(torch.cuda.device_count(), 1)
    world = num_machines * num_gpus_per_machine
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        # already launched by torchrun: one process per GPU exists
        local = int(os.environ.get("LOCAL_RANK", 0))
        dist.init_process_group(backend=backend)
        if backend == "nccl":
            torch.cuda.set_device(local)
        try:
            main_func(*args)
        finally:
            dist.destroy_process_group()
        return
    if world > 1:
        if dist_url in (None, "auto"):
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            dist_url = "tcp://127.0.0.1:%d" % s.getsockname()[1]
            s.close()
        mp.spawn(_worker, nprocs=num_gpus_per_machine,
                 args=(main_func, world, num_gpus_per_machine, machine_rank, dist_url, backend, args), daemon=False)
    else:
        main_func(*args)
