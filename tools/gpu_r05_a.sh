#!/bin/bash
# round 5, first GPU pass: the corrector on the device against the oracle, and what the corrector pass costs the benchmarked tick
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_corrector.py tests/test_gpu_refine.py::test_refined_appended_knot_matches_oracle tests/test_gpu_fixed_dims.py -x -q -s > gpurun_out/r05a_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05a_tests.log
for c in 0 20; do
  for w in 0; do
    timeout 600 python bench.py --walk --no-cpu-baseline --no-latency --corrector-prim-tol $c --corrector-window $w > gpurun_out/r05a_bench_c${c}_w${w}.log 2>&1
  done
done
timeout 600 python bench.py --walk --no-cpu-baseline --no-latency --corrector-prim-tol 0 --refine-appended-knot 3 > gpurun_out/r05a_bench_refine3.log 2>&1
tail -c 600 gpurun_out/r05a_tests.log
