"""Numerics probe (CPU): the parallel-in-time Riccati of the oracle (riccati_legs > 1) against its serial sweep on the
full-dynamics OCP — steps of one iteration, exact gains, and trajectories after a cold solve and warm MPC ticks."""
import sys
import time
import numpy as np

from mpc_benchmark_amd import aligator
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests import _oracle


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


def one_iteration(N, legs, reduced, seed=2):
    lib = _oracle.load()
    out = {}
    for L in (1, legs):
        fp = FullDynamicsProblem(horizon=N, complete_model=not reduced)
        prob = fp.build(with_terminal_constraint=True)
        solver = fp.make_solver(_native_library=lib)
        solver.setNumThreads(L)
        solver.max_iters = 1
        solver.setup(prob)
        rng = np.random.default_rng(seed)
        xs = [fp.space.integrate(fp.x0, 0.02 * rng.standard_normal(fp.space.ndx)) for _ in range(N + 1)]
        us = [10.0 * rng.standard_normal(fp.nu) for _ in range(N)]
        prob.x0_init = xs[0]
        t0 = time.time()
        solver.run(prob, xs, us)
        nat = solver._native
        d = {"t": time.time() - t0}
        for name in ("dx", "du", "dvs", "dlams"):
            d[name] = [nat.debug_get(name, k) for k in range(N + (0 if name == "du" else 1))]
        d["K"] = [nat.debug_get("Kexact" if L > 1 else "K", k) for k in range(N)]
        d["xs"] = np.array(solver.results.xs); d["us"] = np.array(solver.results.us)
        d["K0"] = solver.results.controlFeedbacks()[0]
        out[L] = d
    a, b = out[1], out[legs]
    print("N=%d legs=%d reduced=%s  (serial %.2fs, legs %.2fs)" % (N, legs, reduced, a["t"], b["t"]))
    for name in ("dx", "du", "dvs", "dlams", "K"):
        errs = [rel(y, x) for x, y in zip(a[name], b[name])]
        print("   %-6s max rel err %.3e (knot %d)" % (name, max(errs), int(np.argmax(errs))))
    print("   xs %.3e us %.3e K0 %.3e" % (rel(b["xs"], a["xs"]), rel(b["us"], a["us"]), rel(b["K0"], a["K0"])))


def cold_and_ticks(N, legs, reduced, ticks=10):
    lib = _oracle.load()
    res = {}
    for L in (1, legs):
        fp = FullDynamicsProblem(horizon=N, complete_model=not reduced)
        prob = fp.build()
        solver = fp.make_solver(_native_library=lib)
        solver.setNumThreads(L)
        solver.setup(prob)
        xs, us = fp.initial_guess()
        solver.run(prob, xs, us)
        r = solver.results
        tr = [(np.array(r.xs), np.array(r.us), r.num_iters)]
        solver.max_iters = 1
        xs, us = list(r.xs), list(r.us)
        for t in range(ticks):
            xs = xs[1:] + [xs[-1]]; us = us[1:] + [us[-1]]
            prob.x0_init = xs[0]
            solver.setup(prob)
            solver.run(prob, xs, us)
            xs, us = list(solver.results.xs), list(solver.results.us)
            tr.append((np.array(xs), np.array(us), solver.results.num_iters))
        res[L] = tr
    print("cold solve + %d ticks, N=%d legs=%d reduced=%s: cold iters %d vs %d" % (ticks, N, legs, reduced, res[1][0][2], res[legs][0][2]))
    for t, (a, b) in enumerate(zip(res[1], res[legs])):
        print("   tick %2d xs %.3e us %.3e" % (t, rel(b[0], a[0]), rel(b[1], a[1])))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    legs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    reduced = (sys.argv[3] != "complete") if len(sys.argv) > 3 else True
    one_iteration(N, legs, reduced)
    if len(sys.argv) > 4:
        cold_and_ticks(N, legs, reduced, int(sys.argv[4]))
