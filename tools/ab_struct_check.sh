#!/bin/bash
# structured [A B] products in the Riccati sweep vs the previous library (libmpc_hip_base.so, a build of the last commit): parity tests, then the bench
O=gpurun_out/${1:-abs}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in new base; do
  LIB=""
  [ $v = base ] && LIB="--lib mpc_benchmark_amd/csrc/libmpc_hip_base.so"
  timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-latency $LIB > $O/bench_s1_$v.log 2>&1; echo "$v s1: $(tail -1 $O/bench_s1_$v.log | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["avg_kernel_ms"])')"
  timeout 600 python bench.py --no-cpu-baseline --no-latency $LIB > $O/bench_$v.log 2>&1; echo "$v s4: $(tail -1 $O/bench_$v.log | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["avg_kernel_ms"])')"
done
timeout 300 python tools/phase_timers.py > $O/phase_timers.txt 2>&1; head -20 $O/phase_timers.txt
timeout 900 python tools/config_sweep.py > $O/other_configs.txt 2>&1; cat $O/other_configs.txt
