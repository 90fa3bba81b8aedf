#!/bin/bash
# round 5: the whole GPU suite and the default bench line after the corrector / fixtures work
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05b_gputests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05b_gputests.log
timeout 900 python bench.py > gpurun_out/r05b_bench_default.log 2>&1
tail -n 5 gpurun_out/r05b_gputests.log
