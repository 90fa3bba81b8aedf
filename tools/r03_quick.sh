set -u
mkdir -p gpurun_out/r03a
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc $?" 
tail -15 gpurun_out/r03a/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-latency > gpurun_out/r03a/bench256.log 2>&1; tail -c 1500 gpurun_out/r03a/bench256.log
timeout 300 python bench.py --no-cpu-baseline --no-latency --lib mpc_benchmark_amd/csrc/variants/libmpc_hip_eval512.so > gpurun_out/r03a/bench512.log 2>&1; tail -c 1500 gpurun_out/r03a/bench512.log
