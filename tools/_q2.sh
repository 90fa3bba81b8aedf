for v in evsync base; do
lib=mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so; [ "$v" = base ] && lib=mpc_benchmark_amd/csrc/libmpc_hip.so
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-latency --lib $lib > /tmp/b.json 2>/tmp/b.err; tail -1 /tmp/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
w=d['roofline']['warmup_kernel_ms_per_step_summed_over_shards']
print('$v   value',d['value'],'ms/step',d['ms_per_step'], 'rescues', d['diverged_instance_rescues'], {k:v for k,v in w.items() if v>0.1})" || tail -3 /tmp/b.err
done
