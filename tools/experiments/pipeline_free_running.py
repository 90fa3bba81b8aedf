"""Developer experiment (GPU box): the kinodynamic control pipeline of the HIP library and of the oracle free-running side by side (tests/test_pipeline.py):
per period the deviations of states / torques / forces and the step lengths the two solves accepted."""
import sys
sys.path.insert(0, ".")
import numpy as np
from tests import _oracle
from tests._metrics import rel_cols
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.pipeline import KinodynamicPipeline
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
def mk(lib):
    p = KinodynamicPipeline(KinodynamicProblem(horizon=20), batch=2, library=lib, walk={}, perturb=True, sigma_q=0.005, sigma_v=0.01)
    p.mpc.options.num_threads = 8
    p.mpc.native.set_options(p.mpc.options)
    p.mpc.prepare_schedule(80)
    p.cold_solve()
    return p
po, ph = mk(_oracle.load()), mk(_capi.load_hip_library())
ph.mpc.native.set_state(po.mpc.native.get_state())
for t in range(50):
    sh, so = ph.tick(), po.tick()
    print(t, list(po.contact_state()), "%.2e %.2e %.2e" % (rel_cols(ph.x, po.x, 1e-3), rel_cols(ph.torques, po.torques, 1.0), rel_cols(ph.forces, po.forces, 1.0)),
          "alpha hip", [s.alpha for s in sh], "oracle", [s.alpha for s in so], "iters", [s.num_iters for s in sh], [s.num_iters for s in so])
