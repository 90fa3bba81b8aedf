#!/bin/bash
# Round profile on the GPU box: default bench line, rocprofv3 kernel stats of the same workload, HBM traffic from
# separate PMC passes (no trace domains together with --pmc).  Outputs under gpurun_out/$1/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-latency"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- python3 bench.py $BENCH_ARGS > $OUT/bench_under_rocprof.log 2>&1
python3 tools/trace_summary.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) $OUT/kernel_trace_summary.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pf -- python3 bench.py --steps 3 --warmup 1 --calibration-ticks 4 --no-cpu-baseline --no-latency > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pw -- python3 bench.py --steps 3 --warmup 1 --calibration-ticks 4 --no-cpu-baseline --no-latency > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -o cf -- tools/pmc/pmc_calib > $OUT/cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -o cw -- tools/pmc/pmc_calib > $OUT/cal_write.log 2>&1
python3 tools/pmc/summarize.py $OUT/traffic.json $OUT/pmc_fetch $OUT/pmc_write $OUT/cal_fetch $OUT/cal_write 1073741824 > $OUT/traffic.log 2>&1
tail -n 60 $OUT/traffic.log
# the same two PMC passes for launches over the whole ensemble of 64 (--streams 1): profiles/traffic_b64_*.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s1 -o pf -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_fetch_s1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s1 -o pw -- python3 bench.py --streams 1 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/pmc_write_s1.log 2>&1
python3 tools/pmc/summarize.py $OUT/traffic_streams1.json $OUT/pmc_fetch_s1 $OUT/pmc_write_s1 $OUT/cal_fetch $OUT/cal_write 1073741824 > $OUT/traffic_streams1.log 2>&1
cp $OUT/traffic.json profiles/traffic_b16_n100_complete.json; cp $OUT/traffic_streams1.json profiles/traffic_b64_n100_complete.json  # on the box: the bench lines below quote them
python3 bench.py > $OUT/bench.log 2> $OUT/bench.err
tail -c 3000 $OUT/bench.log
python3 tools/overlap_report.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) 0.08 > $OUT/overlap_report.txt 2>&1
python3 tools/overlap_report.py $(find $OUT/stats -name '*kernel_trace.csv' | head -1) 0.08 chain | tail -20 >> $OUT/overlap_report.txt 2>&1
python3 bench.py --streams 1 --no-cpu-baseline > $OUT/bench_streams1.log 2>&1
python3 bench.py --batch 256 --no-cpu-baseline --no-latency > $OUT/bench_batch256.log 2>&1
python3 tools/phase_timers.py > $OUT/phase_timers.txt 2>&1
tools/pmc/sq_counters.sh $OUT/sq > $OUT/sq_counters.txt 2>&1
# the raw per-dispatch CSVs are large: keep only the summaries
find $OUT -name '*counter_collection.csv' -size +2M -delete
find $OUT -name '*kernel_trace.csv' -size +8M -delete
ls -la $OUT $OUT/stats/* 2>/dev/null | head -40
