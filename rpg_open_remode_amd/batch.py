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


def init(backend, force=False):
    """Joins the process group of a torch.distributed.run launch.  backend "nccl" is RCCL on ROCm; it needs one device per
    rank, so when more ranks than devices were started (two ranks driven onto a one-GPU lease) the control plane -- a barrier
    and one 32-byte gather -- falls back to gloo while every rank still computes on its (shared) GPU.  `force` creates the
    group even for a single rank (exercises RCCL next to librmd_hip.so in one process)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl":
            n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if n_dev >= world:
                kwargs["device_id"] = torch.device("cuda", local_rank)
            else:
                backend = "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def backend_name():
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def sequences_of_rank(n_sequences, rank, world):
    """Sequence ids owned by `rank`: round-robin, so every rank gets floor or ceil of n/world."""
    return [s for s in range(n_sequences) if s % world == rank]


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        if device is not None and device.type == "cuda" and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def gather_throughput(elapsed_s, units, device=None, extra=()):
    """Whole-job aggregate: (max elapsed over ranks, total units over ranks, per-rank list of records).

    One all-gather of a small f64 record per rank -- (elapsed seconds, units) + `extra` (bench.py sends the number of
    update() calls and the converged-seed count: the 4 x f64 record {seconds, pixels, updates, converged} of SURVEY.md 8e).
    Every rank must pass the same number of extra values."""
    record = (float(elapsed_s), float(units)) + tuple(float(v) for v in extra)
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s, units, [record]
    dev = device if device is not None and dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(record, dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    per_rank = [tuple(float(v) for v in t) for t in out]
    return max(r[0] for r in per_rank), sum(r[1] for r in per_rank), per_rank
