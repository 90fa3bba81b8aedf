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


# ---- BASELINE.json's configuration 5 as the driver launches it: 8 ranks x 64 instances ------------------------------------------------------------
def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _cpu_port
    pd = FullDynamicsProblem(horizon=2)
    shards = make_bench_shards(pd, _cpu_port.load(), 64, rank=rank, world=world, legs=1, tick_reuse=False)
    for e in shards:
        e.options.num_threads = 1
        e.native.set_options(e.options)
        e.prepare_schedule(3)
        e.cold_solve(max_iters=1)
        e.step()
    ids, blk = allgather_results(shards, dist)
    if rank == 0:
        out.put((ids, blk[:, :shards[0].dims.nx].copy(), np.concatenate([e.instance_ids for e in shards]), shards[0].x0.copy(), np.isfinite(blk).all()))
    dist.destroy_process_group()


def test_eight_ranks_of_64_instances_are_the_512_instance_ensemble():
    """bench.py's shard code with world_size 8 and 64 instances per rank (512 / 8): every rank holds the instances
    i = rank (mod 8) of ONE rng stream drawn in instance order, the round-end all-gather returns all 512 result blocks ordered by instance id.
    (The CPU port stands in for the device, horizon 2: sharding, seeding and the gather are what is under test.)"""
    from mpc_benchmark_amd.ensemble import ensemble_initial_states
    world = 8
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_worker8, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    ids, x_first, ids0, x0_rank0, finite = out.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert np.array_equal(ids, np.arange(512)) and finite
    assert np.array_equal(ids0, np.arange(0, 512, 8))
    pd = FullDynamicsProblem(horizon=2)
    prob = pd.build(with_terminal_constraint=True)
    stream = ensemble_initial_states(prob.x0_init, prob.stages[0].xspace, 512, 20250304)
    assert np.array_equal(x0_rank0, stream[0::8])          # rank 0 drew nothing of its own: rows 0, 8, 16, ... of the one stream
    assert np.array_equal(stream[0], np.asarray(prob.x0_init))  # instance 0 is the nominal robot
    # knot 0 of every gathered trajectory is the measured state its instance went on from: distinct instances, ordered by id
    assert len({x.tobytes() for x in x_first}) == 512
