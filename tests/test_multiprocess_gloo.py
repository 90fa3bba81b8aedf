"""N > 1 path on CPU: two processes (gloo) run bench.py's own shard construction — ``make_bench_shards``: SURVEY.md §8d config 5
seeding (one rng stream in instance order, instance i on rank i mod G) — advance their shards independently (no data-path
collective) and exchange the result blocks at the end (``allgather_results``).  The gathered union must equal the
single-process ensemble of all instances bit for bit.  The CPU oracle stands in for the device here: the sharding / seeding /
gather logic is what is under test."""
import os

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from mpc_benchmark_amd.ensemble import allgather_results, make_bench_shards, shard_instances
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

PER_RANK, HORIZON, TICKS = 2, 4, 2


def _run(rank, world, per_rank=PER_RANK):
    from tests import _oracle
    pd = FullDynamicsProblem(horizon=HORIZON)
    shards = make_bench_shards(pd, _oracle.load(), per_rank, rank=rank, world=world, legs=1, tick_reuse=False)
    for e in shards:
        e.prepare_schedule(TICKS + 2)
        e.cold_solve(max_iters=3)
        for _ in range(TICKS):
            e.step()
    return shards


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = _run(rank, world)
    ids, blk = allgather_results(shards, dist)
    if rank == 0:
        out.put((ids, blk, np.concatenate([e.instance_ids for e in shards])))
    dist.destroy_process_group()


def test_two_rank_ensemble_equals_the_single_rank_ensemble():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    ids, blk, ids0 = out.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ids, np.arange(2 * PER_RANK))
    assert np.array_equal(ids0, shard_instances(2 * PER_RANK, 0, 2)) and np.array_equal(ids0, [0, 2])
    # the same ensemble on one rank: instance i of the union is instance i here, bit for bit
    ids1, blk1 = allgather_results(_run(0, 1, 2 * PER_RANK))
    assert np.array_equal(ids1, ids)
    assert np.array_equal(blk1, blk)
    # instances differ (randomised initial states), instance 0 is the nominal one
    assert np.max(np.abs(blk[1] - blk[0])) > 1e-6
