for v in base dab tree all3 se3h; do
  lib=mpc_benchmark_amd/csrc/variants/libmpc_hip_$v.so; [ "$v" = base ] && lib=mpc_benchmark_amd/csrc/libmpc_hip.so
  echo "=== $v"; MPC_HIP_LIBRARY=$PWD/$lib timeout 300 python3 tools/experiments/pipeline_free_running.py 2>&1 | awk '{print $1, $4, $5, $6, $7}' | tr '\n' ';' ; echo
done
