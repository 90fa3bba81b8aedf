#!/bin/bash
# usage: bash tools/legs_sweep.sh "<legs list>" "<streams list>" [extra bench args]
for legs in $1; do for st in $2; do
  echo "legs=$legs streams=$st"; timeout 300 python3 bench.py --steps 30 --warmup 3 --legs $legs --streams $st --no-cpu-baseline $3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('   value',d['value'],'ms/step',d['ms_per_step'],'p50',d['p50_ms_per_solve_batch1'],'dom',r['kernel'],r['avg_kernel_ms'],'late',d['config']['late_releases'],'rescues',d['diverged_instance_rescues'])
print('   ',{k:v for k,v in r['warmup_kernel_ms_per_step_summed_over_shards'].items() if v>0.1})
"; done; done
