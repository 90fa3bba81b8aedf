export PHASE_BATCH=64
echo "=== p11 timers"; PHASE_P11=1 python3 tools/phase_timers.py mpc_benchmark_amd/csrc/variants/libmpc_hip_p11.so 2>&1 | grep -E "EVAL"
echo "=== sub timers"; PHASE_SUB=1 python3 tools/phase_timers.py mpc_benchmark_amd/csrc/variants/libmpc_hip_sub.so 2>&1 | grep -E "EVAL"
PARITY=1 PARITY_TESTS="tests/test_gpu_fulldynamic.py tests/test_gpu_fixed_dims.py tests/test_gpu_kinodynamic.py" tools/exp_variants.sh cone hmir all5
