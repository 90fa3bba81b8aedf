#!/bin/bash
# round 5: chol16 on the matrix cores — the library with and without it
mkdir -p gpurun_out
for v in default chol16mfma; do
  LIBARG=""; [ $v != default ] && LIBARG="--lib mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so"
  timeout 600 python bench.py --no-cpu-baseline --no-whole-schedule --steps 40 $LIBARG > gpurun_out/r05c_bench_$v.log 2>&1
done
timeout 300 python tools/batch1_kernel_times.py > gpurun_out/r05c_batch1_kernel_times.txt 2>&1
tail -n 22 gpurun_out/r05c_batch1_kernel_times.txt
