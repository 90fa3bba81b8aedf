#!/bin/bash
# Developer tool: instruction-fetch counters of the solver kernels (separate rocprofv3 --pmc passes, no trace domains).
# usage: tools/pmc/icache_counters.sh OUTDIR [LIB]   (on the GPU box)
OUT=${1:-gpurun_out/icache}
LIB=${2:-mpc_benchmark_amd/csrc/libmpc_hip.so}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-walk --lib $LIB"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_TC_INST_REQ SQC_TC_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o c -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, os
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
grid = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        kname = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[kname][r["Counter_Name"]].append(float(r["Counter_Value"]))
        grid[kname].append(float(r["Grid_Size"]) / max(1.0, float(r["Workgroup_Size"])))
for k in sorted(acc):
    if not any(s in k for s in ("riccati", "eval_multibody", "k_leg_knot", "k_leg_condense")):
        continue
    print(k, " workgroups per full launch:", int(max(grid[k])))
    for c in sorted(acc[k]):
        v = acc[k][c]
        mx = max(v)
        keep = [x for x in v if x >= 0.9 * mx]
        print("   %-32s %16.0f  (mean of %d full launches)  %12.0f per workgroup" % (c, sum(keep) / len(keep), len(keep), sum(keep) / len(keep) / max(grid[k])))
PY
find $OUT -name '*counter_collection.csv' -size +1M -delete
