#!/bin/bash
# Developer tool: SQ instruction-mix / stall counters of the solver kernels (separate rocprofv3 --pmc passes, no trace domains).
# usage: tools/pmc/sq_counters.sh OUTDIR   (on the GPU box)
OUT=${1:-gpurun_out/sq}
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--streams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64"; do
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
    if not any(s in k for s in ("riccati", "eval_multibody", "k_leg", "k_duals", "k_forward", "k_lagrangian")):
        continue
    print(k, " workgroups per full launch:", int(max(grid[k])))
    for c in sorted(acc[k]):
        v = acc[k][c]
        mx = max(v)
        keep = [x for x in v if x >= 0.9 * mx]
        print("   %-32s %16.0f  (mean of %d full launches)  %12.0f per workgroup" % (c, sum(keep) / len(keep), len(keep), sum(keep) / len(keep) / max(grid[k])))
PY
find $OUT -name '*counter_collection.csv' -size +1M -delete
