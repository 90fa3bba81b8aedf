#!/bin/bash
# round 5, step t: the floor under the measured soles (mpc_walk_config.floor_z, ABI 3)
mkdir -p gpurun_out/r05t
timeout 1500 python -m pytest tests/test_walk_generator.py tests/test_abi_library.py "tests/test_gpu_corrector.py::test_whole_schedule_on_one_iteration_per_tick" tests/test_gpu_walk.py -q -m gpu -s 2>&1 | tail -14 | cut -c1-300 > gpurun_out/r05t/tests.log
cat gpurun_out/r05t/tests.log
bash tools/experiments/r05_floor.sh > gpurun_out/r05t/floor_seeds.txt 2>&1; cat gpurun_out/r05t/floor_seeds.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05t/bench_driver_args.log 2>&1
tail -c 200 gpurun_out/r05t/bench_driver_args.log
