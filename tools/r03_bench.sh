# developer helper: default bench (frozen + walk), summary of the JSON line; outputs under gpurun_out/$1
TAG=${1:-r03q}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python bench.py "$@" > gpurun_out/$TAG/bench.log 2> gpurun_out/$TAG/bench.err || tail -20 gpurun_out/$TAG/bench.err
python - <<PY
import json
d = json.loads(open('gpurun_out/$TAG/bench.log').read().strip().split('\n')[-1])
print('solves/s', d['value'], 'ms/tick', d['ms_per_step'], '|', d.get('tick_mode'), '| dominant', d['roofline']['kernel'], d['roofline']['avg_kernel_ms'], 'frac', d['roofline']['frac'])
print(json.dumps(d.get('measurements'), indent=1))
print('rescues', d['diverged_instance_rescues'], 'no-step', d['instance_ticks_without_step'], 'p50', d.get('p50_ms_per_solve_batch1'), 'cpu', d.get('cpu_baseline'))
PY
