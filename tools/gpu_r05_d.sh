#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05d_gputests.log 2>&1
echo "tests rc $?" >> gpurun_out/r05d_gputests.log
tail -n 6 gpurun_out/r05d_gputests.log
bash tools/gpu_profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
tail -n 5 gpurun_out/r05_profile_round.log
