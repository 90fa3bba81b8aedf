"""bench.py's contract with the driver, as far as it can be held without a GPU: the command line parses (``--help``, the driver's flags), and the line the
last GPU run of the round printed (profiles/r05_bench_driver_args.log, committed with the round) has every key the contract names — so that a change of
the JSON layout shows up here and not in the driver's parser."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(path):
    with open(path) as f:
        rows = [l for l in f if l.startswith("{")]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    return json.loads(rows[0])


def test_command_line_parses():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--no-floor", "--corrector-prim-tol", "--refine-appended-knot", "--walk-generator", "--no-whole-schedule"):
        assert flag in out.stdout, flag


def test_committed_line_has_the_contract_keys():
    d = _line(os.path.join(ROOT, "profiles", "r05_bench_driver_args.log"))
    assert d["metric"] == "mpc_solves_per_sec" and d["unit"] == "solves/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]          # value = instances per tick / time per tick
    assert "workload" in d["config"] and "model" not in d["config"] and "parallelism" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or (r["traffic"] > 0 and "traffic_source" in r)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "solves/s" and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) > 20
    w = d["whole_schedule"]
    assert w["ticks"] == 999 and w["instance_losses"] == 0 and not w["nominal_instance_lost"] and set(w["ms_per_tick"]) >= {"mean", "p50", "p90", "p95", "max"}
    for k in ("whole_schedule_plain_warm_start", "whole_schedule_exact_iteration_budget"):
        assert d[k]["ticks"] == 999 and "instance_losses" in d[k]
