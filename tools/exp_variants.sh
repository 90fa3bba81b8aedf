#!/bin/bash
# developer experiment: kernel tuning variants (built under mpc_benchmark_amd/csrc/variants/) on the GPU box
OUT=gpurun_out/exp; mkdir -p $OUT
B="--steps 8 --warmup 2 --no-cpu-baseline --no-latency"
for v in "$@"; do
  lib=mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so
  [ "$v" = base ] && lib=mpc_benchmark_amd/csrc/libmpc_hip.so
  echo "=== $v"
  python3 bench.py $B --lib $lib 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], 'solves/s', j['ms_per_step'], 'ms/step', j['roofline']['warmup_kernel_ms_per_step_summed_over_shards'])
    else: print(l, end='')
"
  python3 tools/phase_timers.py $lib 2>&1 | grep -E "EVAL|total" 
done
