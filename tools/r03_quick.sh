# developer helper: GPU parity tests + a short default bench + the stage kernel's phase timers; outputs under gpurun_out/$1
set -u
TAG=${1:-r03q}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/$TAG/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-latency > gpurun_out/$TAG/bench.log 2>&1
python - <<PY
import json
d = json.loads(open('gpurun_out/$TAG/bench.log').read().strip().split('\n')[-1])
print('solves/s', d['value'], 'ms/tick', d['ms_per_step'], 'dominant', d['roofline']['kernel'], d['roofline']['avg_kernel_ms'])
print(d['roofline']['warmup_kernel_ms_per_step_summed_over_shards'])
PY
python tools/phase_timers.py > gpurun_out/$TAG/phase.txt 2>&1; grep EVAL gpurun_out/$TAG/phase.txt
