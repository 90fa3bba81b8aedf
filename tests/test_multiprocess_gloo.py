"""N > 1 path on CPU: two processes (gloo) each advance their own shard of the ensemble — instances are independent,
there is no data-path collective — then gather the per-rank results.  The CPU oracle stands in for the device here
(the sharding / seeding / reduction logic of bench.py is what is under test)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _oracle
    ens = EnsembleMPC(CentroidalProblem(horizon=10), batch=2, library=_oracle.load(), seed=100 + rank)
    ens.x0[1, :3] += 0.01 * (rank + 1)  # rank-dependent initial state of the second instance
    ens.prepare_schedule(6)
    ens.cold_solve(max_iters=20)
    for _ in range(3):
        ens.step()
    us = torch.from_numpy(ens.results(gains=False)["us"].copy())
    gathered = [torch.zeros_like(us) for _ in range(world)]
    dist.all_gather(gathered, us)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put((np.stack([g.numpy() for g in gathered]), float(t.item())))
    dist.destroy_process_group()


def test_two_rank_ensemble_shards():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res, tmax = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0 and res.shape[0] == 2
    # instance 0 is identical on both ranks (same x0), instance 1 differs (rank-dependent x0)
    assert np.allclose(res[0][0], res[1][0])
    assert np.max(np.abs(res[0][1] - res[1][1])) > 1e-6
    # and each shard equals a single-process run with the same data
    from tests import _oracle
    ens = EnsembleMPC(CentroidalProblem(horizon=10), batch=2, library=_oracle.load(), seed=101)
    ens.x0[1, :3] += 0.02
    ens.prepare_schedule(6)
    ens.cold_solve(max_iters=20)
    for _ in range(3):
        ens.step()
    assert np.allclose(ens.results(gains=False)["us"], res[1], rtol=1e-12, atol=1e-12)
