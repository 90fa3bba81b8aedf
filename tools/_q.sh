python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > /tmp/b.json 2>/dev/null; tail -1 /tmp/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
w=d['roofline']['warmup_kernel_ms_per_step_summed_over_shards']
print('   value',d['value'],'ms/step',d['ms_per_step'],'p50',d['p50_ms_per_solve_batch1'], 'rescues', d['diverged_instance_rescues'], {k:v for k,v in w.items() if v>0.1})"
python tools/batch1_kernel_times.py 16 2>&1 | grep -E "eval_stage|sum of"
