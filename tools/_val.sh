timeout 1500 python -m pytest tests/test_gpu_legs.py tests/test_gpu_fixed_dims.py tests/test_gpu_bench_config.py -m gpu -q -x 2>&1 | tail -3
python tools/latency_vs_legs.py 2>&1 | grep "legs 32"
python tools/batch1_kernel_times.py 32 2>&1 | head -3
python tools/tree_fallbacks.py 2>&1 | tail -3
