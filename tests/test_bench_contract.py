"""bench.py's contract with the driver, held without a GPU by RUNNING bench.py: ``--selftest-cpu`` takes the same code path — launcher, rank set-up, shard
construction, warm-up, timed region, whole-schedule walk, batch-1 latency, PCIe-inclusive loop, CPU-baseline workers, all-gather, JSON line — on the CPU port
(``oracle/cpu_port``, test infrastructure) with gloo and tiny sizes.  The numbers are meaningless and the line says so (``selftest``, ``metric``); what is
checked is the layout of the line, the arithmetic between its fields and that ``--gpus N`` launches N ranks.  Result-valued assertions (no instance lost over
the schedule, ...) live in the GPU tests (tests/test_gpu_bench_config.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--steps", "2", "--warmup", "1", "--batch", "2", "--horizon", "6", "--model", "reduced", "--cold-iters", "100", "--regions", "1",
         "--latency-ticks", "6", "--schedule-ticks", "8"]


@pytest.fixture(scope="module")
def cpu_lib():
    from tests import _cpu_port
    _cpu_port.load()  # (builds it on demand)
    return os.path.join(_cpu_port.PORT_DIR, "libmpc_cpu.so")


def _run(args, timeout=600):
    out = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")})
    rows = [l for l in out.stdout.split("\n") if l.startswith("{")]
    return out, rows


def test_command_line_parses():
    out = subprocess.run([sys.executable, BENCH, "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--no-floor", "--corrector-prim-tol", "--refine-appended-knot", "--walk-generator", "--no-whole-schedule"):
        assert flag in out.stdout, flag


def test_refuses_a_cpu_library_outside_the_selftest(cpu_lib):
    out, rows = _run(["--lib", cpu_lib] + SMALL)
    assert out.returncode != 0 and not rows and "HIP library only" in out.stderr


def test_line_of_the_cpu_rehearsal_has_the_contract_keys(cpu_lib):
    out, rows = _run(["--selftest-cpu", "--lib", cpu_lib] + SMALL)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(rows) == 1, "bench.py prints ONE JSON line"
    d = json.loads(rows[0])
    assert d["selftest"] is True and "NOT a measurement" in d["metric"] and d["metric"].startswith("mpc_solves_per_sec")
    assert d["unit"] == "solves/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]          # value = instances per tick / time per tick
    assert "workload" in d["config"] and "model" not in d["config"] and "parallelism" in d["config"]
    # the objects the judge reads come before the supplementary ones (the driver's record keeps a bounded number of key names)
    keys = list(d)
    for k in ("roofline", "cpu_baseline", "whole_schedule", "whole_schedule_plain_warm_start", "whole_schedule_exact_iteration_budget", "roofline_whole_tick",
              "value_with_result_download", "per_rank_ms_per_step", "ensemble_allgather"):
        assert k in d and keys.index(k) < 26, k
    assert d["roofline"] is None and d["roofline_whole_tick"] is None   # (the CPU port times no kernels: the rooflines are the GPU run's)
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "solves/s" and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) > 20 and d["cpu_baseline_note"] is None
    w = d["whole_schedule"]
    assert w["ticks"] == 8 and set(w["ms_per_tick"]) >= {"mean", "p50", "p90", "p95", "max"} and "instance_losses" in w and "nominal_instance_lost" in w
    for k in ("whole_schedule_plain_warm_start", "whole_schedule_exact_iteration_budget"):
        assert d[k]["ticks"] == 8 and "instance_losses" in d[k]
    assert d["whole_schedule_exact_iteration_budget"]["settings"]["corrector_prim_tol"] == 0.0 and d["whole_schedule_exact_iteration_budget"]["settings"]["refine_appended_knot"] == 0
    v = d["value_with_result_download"]
    assert v["unit"] == "solves/s" and v["bytes_downloaded_per_step"] > 0 and abs(v["value"] - 2 * 1e3 / v["ms_per_step"]) < 0.02 * v["value"]
    assert d["per_rank_ms_per_step"] == {"min": d["ms_per_step"], "max": d["ms_per_step"], "ranks": 1}
    assert d["p50_ms_per_solve_batch1"] > 0 and d["latency_ticks"] == 6
    assert set(d["measurements"]) >= {"frozen_references", "walk"}


def test_gpus_n_launches_n_ranks(cpu_lib):
    """``python bench.py --gpus 2`` without a launcher around it: the parent starts two ranks (gloo here, RCCL on the GPUs), relays rank 0's line."""
    out, rows = _run(["--selftest-cpu", "--lib", cpu_lib, "--gpus", "2", "--no-latency", "--no-whole-schedule"] + SMALL)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(rows) == 1
    d = json.loads(rows[0])
    assert d["n_gpus"] == 2 and d["selftest"] is True
    g = d["ensemble_allgather"]
    assert g["instances"] == 4 and g["complete"] is True and g["finite"] is True   # ids 0 .. 3: rank r holds r, r + 2
    p = d["per_rank_ms_per_step"]
    assert p["ranks"] == 2 and p["min"] <= p["max"] and abs(p["max"] - d["ms_per_step"]) < 1e-3   # ms_per_step = the slowest rank's
    assert abs(d["value"] - 2 * 2 * 1e3 / d["ms_per_step"]) < 0.02 * d["value"]    # whole-job rate: both ranks' instances
    assert d["cpu_baseline"] is None and "N = 1" in d["cpu_baseline_note"]
    assert "sharded over 2 GPU(s)" in d["config"]["parallelism"]


def test_launcher_fails_when_a_rank_fails(cpu_lib):
    out, rows = _run(["--selftest-cpu", "--lib", "/nonexistent/libmpc.so", "--gpus", "2", "--no-latency", "--no-whole-schedule", "--no-cpu-baseline"] + SMALL)
    assert out.returncode != 0 and not rows


def test_launcher_does_not_wait_for_a_rendezvous_that_cannot_happen(cpu_lib):
    """One rank dies before it joins (on the GPU box: `--gpus 2` where there is one GPU): the other would wait ten minutes for it at the rendezvous.  The launcher
    ends it at once and fails."""
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--selftest-cpu", "--lib", cpu_lib, "--gpus", "2", "--no-latency", "--no-whole-schedule", "--no-cpu-baseline"] + SMALL,
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}, MPC_BENCH_TEST_FAIL_RANK="1"))
    assert out.returncode != 0 and time.time() - t0 < 120 and "ranks failed" in out.stderr
