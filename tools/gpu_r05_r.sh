#!/bin/bash
# round 5, step r: the compressed-[A B] plan of the sweep (make_ric_lds plan 3) for the problems whose plan 1 has no room for the overlapped
# factorisation of Ruu (m = 44 / 34 / 22): the whole GPU suite, then the times
mkdir -p gpurun_out/r05r
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/r05r/tests.log
cat gpurun_out/r05r/tests.log
python tools/kino_tick.py 60 > gpurun_out/r05r/kino_tick.txt 2>&1; head -1 gpurun_out/r05r/kino_tick.txt; tail -9 gpurun_out/r05r/kino_tick.txt
python tools/pipeline_tick.py > gpurun_out/r05r/pipeline_tick.txt 2>&1; COMPLETE=1 python tools/pipeline_tick.py >> gpurun_out/r05r/pipeline_tick.txt 2>&1; cat gpurun_out/r05r/pipeline_tick.txt
python tools/shim_tick_time.py 2>&1 | grep -E 'p50' > gpurun_out/r05r/drop_in_tick.txt; cat gpurun_out/r05r/drop_in_tick.txt
python tools/config_sweep.py > gpurun_out/r05r/other_configs.txt 2>&1; cat gpurun_out/r05r/other_configs.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-whole-schedule > gpurun_out/r05r/bench.log 2>&1; tail -c 600 gpurun_out/r05r/bench.log | head -c 300
