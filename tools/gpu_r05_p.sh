#!/bin/bash
# round 5, step p: the low-level loop of the kinodynamic pipeline inside the library (mpc_qp_low_level_steps)
mkdir -p gpurun_out/r05p
timeout 900 python -m pytest tests/test_pipeline.py tests/test_gpu_qp.py tests/test_abi_library.py -q -m gpu -s -x 2>&1 | tail -15 > gpurun_out/r05p/tests.log
cat gpurun_out/r05p/tests.log
timeout 600 python tools/pipeline_tick.py > gpurun_out/r05p/pipeline_tick.txt 2>&1
COMPLETE=1 timeout 600 python tools/pipeline_tick.py >> gpurun_out/r05p/pipeline_tick.txt 2>&1
cat gpurun_out/r05p/pipeline_tick.txt
