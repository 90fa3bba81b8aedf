# developer helper: GPU parity tests + a short default bench + the stage kernel's phase timers; outputs under gpurun_out/$1
set -u
TAG=${1:-r03q}
mkdir -p gpurun_out/$TAG
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/$TAG/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-latency > gpurun_out/$TAG/bench.log 2> gpurun_out/$TAG/bench.err || tail -20 gpurun_out/$TAG/bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/$TAG/bench.log').read().strip().split('\n')[-1])
print('solves/s', d['value'], 'ms/tick', d['ms_per_step'], d.get('tick_mode'), 'dominant', d['roofline']['kernel'], d['roofline']['avg_kernel_ms'])
print(json.dumps(d.get('measurements'), indent=1))
PY
python tools/phase_timers.py > gpurun_out/$TAG/phase.txt 2>&1; grep EVAL gpurun_out/$TAG/phase.txt | tail -3
