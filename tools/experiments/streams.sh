cd /root/repo
mkdir -p gpurun_out/streams
timeout 300 python tools/shim_tick_time.py 2>&1 | head -3
for st in 1 2 4; do
  timeout 600 python bench.py --no-cpu-baseline --no-latency --no-walk --streams $st > gpurun_out/streams/s$st.log 2> gpurun_out/streams/s$st.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/streams/s$st.log').read().strip().split('\n')[-1])
print('streams', $st, 'solves/s', d['value'], 'ms/tick', d['ms_per_step'], 'period', d['config'].get('shard_period_ms'), 'late', d['config'].get('late_releases'))
PY
done
