#!/bin/bash
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r05o_gputests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05o_gputests.log
tail -n 4 gpurun_out/r05o_gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05o_bench_driver_args.log 2>&1
timeout 900 python bench.py > gpurun_out/r05o_bench.log 2>&1
MPC_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py --no-cpu-baseline --no-latency --no-whole-schedule > gpurun_out/r05o_bench_force_dist.log 2>&1
tail -c 400 gpurun_out/r05o_bench_force_dist.log
