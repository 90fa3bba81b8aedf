#!/bin/bash
# developer experiment: kernel tuning variants (built by tools/build_variant.sh under mpc_benchmark_amd/csrc/variants/) on the GPU box:
#   tools/exp_variants.sh base NAME ...     per variant: [PARITY=1: the stage-kernel / sweep parity tests against the oracle,] a short bench line with the
#   per-kernel times of a tick, the in-kernel phase timers (PHASE_BATCH instances, default 64 = the benchmarked ensemble's contention)
OUT=gpurun_out/exp; mkdir -p $OUT
B="--steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-whole-schedule"
export PHASE_BATCH=${PHASE_BATCH:-64}
for v in "$@"; do
  lib=mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so
  [ "$v" = base ] && lib=mpc_benchmark_amd/csrc/libmpc_hip.so
  echo "=== $v"
  if [ -n "$PARITY" ]; then MPC_HIP_LIBRARY=$PWD/$lib timeout 900 python3 -m pytest -x -q -m gpu ${PARITY_TESTS:-tests/test_gpu_fulldynamic.py tests/test_gpu_fixed_dims.py} 2>&1 | tail -n 3; fi
  timeout 600 python3 bench.py $B --lib $lib 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], 'solves/s', j['ms_per_step'], 'ms/step', json.dumps(j['roofline'].get('warmup_kernel_ms_per_step_summed_over_shards')))
    else: print(l, end='')
"
  timeout 300 python3 tools/phase_timers.py $lib 2>&1 | grep -E "${PHASE_GREP:-EVAL|total}"
done
