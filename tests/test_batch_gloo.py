"""Multi-GPU path on CPU: world_size-2 gloo run of the sharding + throughput gather used by bench.py (batch.py).
The data path has no collective (independent sequences per rank); only the barrier and the final gather communicate."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_sequences, out_dir):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import json
    import time
    import numpy as np
    import torch
    from rpg_open_remode_amd import batch, synth
    import oracles as O

    r, lr, w = batch.init("gloo")
    assert (r, w) == (rank, world)
    mine = batch.sequences_of_rank(n_sequences, r, w)
    # each rank really runs its own independent sequences (CPU oracle as the stand-in for the device path)
    W, H, frames = 64, 48, 4
    batch.barrier()
    t0 = time.perf_counter()
    checks = {}
    for sid in mine:
        seq = synth.Sequence(W, H, frames, seed=sid)
        s = O.Seeds(O.OracleLib("port", 5), W, H, seq.K)
        s.set_reference(seq.images[0], seq.T_curr_world[0], seq.min_depth, seq.max_depth)
        for k in range(1, frames):
            s.update(seq.images[k], seq.T_curr_world[k])
        checks[sid] = float(np.nansum(s.download(O.PLANE_MU)))
    batch.barrier()
    elapsed = time.perf_counter() - t0 + 0.01 * (r + 1)  # distinct per rank so MAX is observable
    units = float(W * H * (frames - 1) * len(mine))
    max_e, total_u, per_rank = batch.gather_throughput(elapsed, units, torch.device("cpu"), extra=(float(len(mine) * (frames - 1)), float(r + 7)))
    json.dump({"rank": r, "mine": mine, "elapsed": elapsed, "max_e": max_e, "total_u": total_u, "per_rank": per_rank, "checks": checks},
              open(os.path.join(out_dir, f"rank{r}.json"), "w"))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.parametrize("n_sequences", [2, 5])
def test_two_rank_gloo_sharding_and_gather(tmp_path, n_sequences):
    import json
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_sequences, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    # every sequence is owned by exactly one rank
    owned = sorted(res[0]["mine"] + res[1]["mine"])
    assert owned == list(range(n_sequences))
    assert abs(len(res[0]["mine"]) - len(res[1]["mine"])) <= 1
    # both ranks agree on the aggregate: MAX of elapsed, SUM of units
    for r in res:
        assert r["max_e"] == pytest.approx(max(res[0]["elapsed"], res[1]["elapsed"]))
        assert r["total_u"] == 64 * 48 * 3 * n_sequences
        assert [rec[1] for rec in r["per_rank"]] == [64 * 48 * 3 * len(res[0]["mine"]), 64 * 48 * 3 * len(res[1]["mine"])]
        # the 4 x f64 record of SURVEY.md 8e: seconds, pixels, update() calls, converged seeds (here: a per-rank marker)
        assert [rec[2:] for rec in r["per_rank"]] == [[3.0 * len(res[0]["mine"]), 7.0], [3.0 * len(res[1]["mine"]), 8.0]]
    # different seeds really are different workloads
    allc = {**res[0]["checks"], **res[1]["checks"]}
    assert len(set(round(v, 3) for v in allc.values())) == n_sequences


def test_single_process_defaults():
    sys.path.insert(0, ROOT)
    from rpg_open_remode_amd import batch
    assert batch.sequences_of_rank(8, 3, 8) == [3]
    assert batch.sequences_of_rank(3, 1, 2) == [1]
    e, u, per = batch.gather_throughput(1.5, 100.0)
    assert (e, u, per) == (1.5, 100.0, [(1.5, 100.0)])
