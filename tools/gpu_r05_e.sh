#!/bin/bash
mkdir -p gpurun_out
bash tools/experiments/r05_sweep3.sh > gpurun_out/r05_sweep3.txt 2>&1
cat gpurun_out/r05_sweep3.txt
bash tools/pmc/sq_counters.sh gpurun_out/r05_sq > gpurun_out/r05_sq_counters.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05e_bench.log 2>&1
grep -o '"p50_ms_per_solve_batch1": [0-9.]*' gpurun_out/r05e_bench.log
