#!/bin/bash
# kernel variant check: [parity tests,] the bench on one stream and the phase timers
O=gpurun_out/${1:-abs}; mkdir -p $O
if [ "$2" != "notest" ]; then timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log; fi
timeout 600 python bench.py --streams 1 --no-cpu-baseline --no-latency > $O/bench_s1_new.log 2>&1; echo "s1: $(tail -1 $O/bench_s1_new.log | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["roofline"]["avg_kernel_ms"])')"
timeout 300 python tools/phase_timers.py > $O/phase_timers.txt 2>&1; head -25 $O/phase_timers.txt | tail -14
