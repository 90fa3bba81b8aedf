#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05g_gputests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05g_gputests.log
tail -n 4 gpurun_out/r05g_gputests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05g_bench_driver_args.log 2>&1
timeout 900 python bench.py > gpurun_out/r05g_bench.log 2>&1
