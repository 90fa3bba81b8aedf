#!/bin/bash
# round 5: odd leading dimensions for W / CT / VX and Y in the sweep's LDS carve-out — parity, time, bank conflicts
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/mpc_benchmark_amd/csrc/variants/libmpc_hip_ricpad.so
MPC_HIP_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_fulldynamic.py tests/test_gpu_legs.py tests/test_gpu_kinodynamic.py -q -x > gpurun_out/r05f_tests_ricpad.log 2>&1
tail -n 3 gpurun_out/r05f_tests_ricpad.log
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0"
for v in default ricpad; do
  LIBARG=""; [ $v != default ] && LIBARG="--lib $V"
  timeout 300 python bench.py $ARGS $LIBARG > gpurun_out/r05f_bench_$v.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/r05f_pmc_$v -o c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 $LIBARG > gpurun_out/r05f_pmc_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, json
from collections import defaultdict
for v in ("default", "ricpad"):
    d = [json.loads(l) for l in open("gpurun_out/r05f_bench_%s.log" % v) if l.startswith("{")][0]
    print(v, "frozen", d["value"], "sweep", d["roofline"]["avg_kernel_ms"], d["roofline"]["warmup_kernel_ms_per_step_summed_over_shards"])
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob("gpurun_out/r05f_pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f, newline="")):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        if "riccati" in k or "k_leg_knot" in k:
            row = {c: max(vs) for c, vs in acc[k].items()}
            print("   ", k[:60], {c: int(x) for c, x in row.items()}, "conflict share %.3f" % (row.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, row.get("SQ_LDS_IDX_ACTIVE", 1))))
PY
find gpurun_out -name '*counter_collection.csv' -size +1M -delete
