"""Developer tool: wall time of ONE MPC tick through the Python mirror (the drop-in path of the reference scripts):
reference generation + 2 N setReference + replaceStageCircular + terminal-constraint rebuild + setup + run at N = 100,
batch = 1 — host work included, unlike bench.py's device-resident ensemble loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.walking_loop import WalkingMPCLoop

fp = FullDynamicsProblem(horizon=100, complete_model=True)
solver = fp.make_solver()
loop = WalkingMPCLoop(fp, solver, start_tick=0)
for _ in range(5):
    loop.tick()
tt, tr = [], []
for _ in range(40):
    t0 = time.perf_counter()
    loop.tick()
    tt.append((time.perf_counter() - t0) * 1e3)
import cProfile, pstats
print("p50 %.2f ms  p90 %.2f ms per tick through the aligator mirror (N=100, complete model, batch 1, riccati_legs %d)" % (
    np.percentile(tt, 50), np.percentile(tt, 90), solver._legs()))
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    loop.tick()
pr.disable()
print("profile of 40 ticks (times are totals: divide by 40):")
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
