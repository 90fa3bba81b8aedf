python -m pytest tests/test_checkpoint.py tests/test_pipeline.py tests/test_qp_utils.py tests/test_simulate_push.py tests/test_walk_generator.py tests/test_gpu_qp.py -x -q -m gpu 2>&1 | tail -n 6
export PHASE_BATCH=64
echo "=== sub2 (default + sub timers)"; PHASE_SUB=1 python3 tools/phase_timers.py mpc_benchmark_amd/csrc/variants/libmpc_hip_sub2.so 2>&1 | grep -E "EVAL" | grep -E "sub:|chol M|total"
echo "=== cholnsub"; PHASE_SUB=1 python3 tools/phase_timers.py mpc_benchmark_amd/csrc/variants/libmpc_hip_cholnsub.so 2>&1 | grep -E "EVAL" | grep -E "sub:|chol M|total"
echo "=== splitp"; PHASE_P11=1 python3 tools/phase_timers.py mpc_benchmark_amd/csrc/variants/libmpc_hip_splitp.so 2>&1 | grep -E "EVAL" | grep -E "P11|R1 in|contact rows|total"
PARITY=1 PARITY_TESTS="tests/test_gpu_fulldynamic.py tests/test_gpu_fixed_dims.py tests/test_gpu_kinodynamic.py" tools/exp_variants.sh base split
