for v in dab all3 se3; do
  lib=mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so
  echo "=== $v"; MPC_HIP_LIBRARY=$PWD/$lib timeout 300 python3 tools/experiments/pipeline_one_step.py 2>&1 | awk '{print $1, $12, $14}' | tr '\n' ';' ; echo
done
