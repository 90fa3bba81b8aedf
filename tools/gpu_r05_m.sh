#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_walk_generator.py tests/test_abi_library.py -m gpu -q -s > gpurun_out/r05m_tests.log 2>&1
tail -n 6 gpurun_out/r05m_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-latency --walk --walk-generator device > gpurun_out/r05m_bench_device_gen.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-latency --walk --no-whole-schedule --steps 20 --warmup 5 > gpurun_out/r05m_bench_host_gen.log 2>&1
python3 - <<'PY'
import json
for f in ("gpurun_out/r05m_bench_device_gen.log", "gpurun_out/r05m_bench_host_gen.log"):
    try:
        d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    except Exception as e:
        print(f, "no line"); print(open(f).read()[-1500:]); continue
    print(f, {k: (v["value"], v["ms_per_step"]) for k, v in d["measurements"].items()})
    w = d.get("whole_schedule")
    if w: print("   whole", w["solves_per_sec"], w["ms_per_tick"], "lost", w["instance_losses"], "corr", w["corrector_ticks"])
PY
