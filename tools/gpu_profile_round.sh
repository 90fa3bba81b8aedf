#!/bin/bash
# Round profile on the GPU box: default bench line, rocprofv3 kernel stats of the same workload, HBM traffic from
# separate PMC passes (no trace domains together with --pmc).  Outputs under gpurun_out/$1/; a step that fails stops the
# script BEFORE anything is copied over the committed profiles/ files.
set -u
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
step() { echo "== $*" >&2; "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "FAILED (rc $rc): $*" >&2; exit $rc; fi; }
BENCH_ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 --cold-iters 100"  # (frozen references: every candidate launch is the <3> kernel; corrector off: its passes are launches most workgroups sit out, they would dilute the per-kernel averages; the scripts' 100 cold iterations: the tail of a longer cold solve runs with a handful of active instances and would pull the per-kernel averages down)
step rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 bench.py $BENCH_ARGS > $OUT/bench_under_rocprof.log 2>&1
TRACE=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
[ -n "$TRACE" ] || { echo "no kernel trace" >&2; exit 1; }
step python3 tools/trace_summary.py $TRACE $OUT/kernel_trace_summary.csv
# PMC passes: the whole ensemble per launch (default bench: one lock-step ensemble) and one shard of 32 per launch (--streams 2)
step rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 > $OUT/pmc_fetch.log 2>&1
step rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pw -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 > $OUT/pmc_write.log 2>&1
step rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cf -- tools/pmc/pmc_calib > $OUT/cal_fetch.log 2>&1
step rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cw -- tools/pmc/pmc_calib > $OUT/cal_write.log 2>&1
python3 tools/pmc/summarize.py $OUT/traffic.json $OUT/pmc_fetch $OUT/pmc_write $OUT/cal_fetch $OUT/cal_write 1073741824 > $OUT/traffic.log 2>&1 || { echo "summarize failed" >&2; exit 1; }
tail -n 40 $OUT/traffic.log
step rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s2 -o pf -- python3 bench.py --streams 2 --steps 3 --warmup 1 --calibration-ticks 4 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 > $OUT/pmc_fetch_s2.log 2>&1
step rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s2 -o pw -- python3 bench.py --streams 2 --steps 3 --warmup 1 --calibration-ticks 4 --no-cpu-baseline --no-latency --no-walk --no-whole-schedule --corrector-prim-tol 0 > $OUT/pmc_write_s2.log 2>&1
python3 tools/pmc/summarize.py $OUT/traffic_streams2.json $OUT/pmc_fetch_s2 $OUT/pmc_write_s2 $OUT/cal_fetch $OUT/cal_write 1073741824 > $OUT/traffic_streams2.log 2>&1 || { echo "summarize failed" >&2; exit 1; }
# on the box: the bench lines below quote them (one launch of the default bench serves all 64 instances)
cp $OUT/traffic.json profiles/traffic_b64_n100_complete.json; cp $OUT/traffic_streams2.json profiles/traffic_b32_n100_complete.json
python3 bench.py > $OUT/bench.log 2> $OUT/bench.err || { echo "bench failed" >&2; tail -5 $OUT/bench.err; exit 1; }
tail -c 3500 $OUT/bench.log
python3 tools/overlap_report.py $TRACE 0.08 > $OUT/overlap_report.txt 2>&1
python3 tools/overlap_report.py $TRACE 0.08 chain | tail -24 >> $OUT/overlap_report.txt 2>&1
python3 bench.py --streams 2 --no-cpu-baseline --no-latency --no-whole-schedule > $OUT/bench_streams2.log 2>&1
python3 bench.py --legs 8 --no-cpu-baseline --no-latency --no-whole-schedule > $OUT/bench_legs8.log 2>&1
python3 bench.py --batch 256 --no-cpu-baseline --no-latency --no-whole-schedule > $OUT/bench_batch256.log 2>&1
python3 tools/phase_timers.py > $OUT/phase_timers.txt 2>&1
MPC_LEGS_CHAIN=1 python3 tools/legs_phase_timers.py 4 > $OUT/legs_phase_timers.txt 2>&1  # (the chain over the cuts: its kernel carries the phase timers of the elimination)
python3 tools/config_sweep.py > $OUT/other_configs.txt 2>&1
python3 tools/shim_tick_time.py 2>&1 | grep -E 'p50' > $OUT/drop_in_tick.txt
python3 tools/latency_vs_legs.py > $OUT/latency_vs_legs.txt 2>&1
python3 tools/kino_tick.py 60 > $OUT/kino_tick.txt 2>&1
python3 tools/qp_bench.py > $OUT/qp_bench.txt 2>&1
python3 tools/batch1_kernel_times.py 32 > $OUT/batch1_kernel_times.txt 2>&1
python3 tools/occupancy_report.py > $OUT/occupancy_kinodynamic.txt 2>&1
python3 tools/occupancy_report.py --problem full --batches 16,64,256 > $OUT/occupancy_fulldynamic.txt 2>&1
tools/pmc/sq_counters.sh $OUT/sq > $OUT/sq_counters.txt 2>&1
PHASE_BATCH=64 python3 tools/phase_timers.py > $OUT/phase_timers_batch64.txt 2>&1
python3 tools/legs_phase_timers.py 4 64 > $OUT/legs_phase_timers_batch64.txt 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.log 2> $OUT/bench_driver_args.err
# the raw per-dispatch CSVs are large: keep only the summaries
find $OUT -name '*counter_collection.csv' -size +2M -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
ls -la $OUT | head -40
