"""Batched mode: independent image sequences sharded one-per-GPU (BASELINE.json config 4).

The depth filter has no cross-sequence state, so the path shards perfectly across sequences and needs
no collective on the data path.  The only communication is the throughput gather at the end (RCCL when
the backend is "nccl", gloo in the CPU tests): a barrier on both sides of the timed region, MAX of the
per-rank elapsed time and SUM of the per-rank work.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) if absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def sequences_of_rank(n_sequences, rank, world):
    """Sequence ids owned by `rank`: round-robin, so every rank gets floor or ceil of n/world."""
    return [s for s in range(n_sequences) if s % world == rank]


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def gather_throughput(elapsed_s, units, device=None):
    """Whole-job aggregate: (max elapsed over ranks, total units over ranks, per-rank list of (elapsed, units))."""
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s, units, [(elapsed_s, units)]
    dev = device if device is not None else torch.device("cpu")
    mine = torch.tensor([float(elapsed_s), float(units)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    per_rank = [(float(t[0]), float(t[1])) for t in out]
    return max(e for e, _ in per_rank), sum(u for _, u in per_rank), per_rank
