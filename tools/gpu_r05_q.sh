#!/bin/bash
# round 5, final run: GPU suite, smoke, bench with the driver's arguments and the default one
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r05q_gputests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05q_gputests.log
tail -n 4 gpurun_out/r05q_gputests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05q_smoke.log 2>&1; tail -n 2 gpurun_out/r05q_smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05q_bench_driver_args.log 2>&1
timeout 900 python bench.py > gpurun_out/r05q_bench.log 2>&1
tail -c 300 gpurun_out/r05q_bench.log
