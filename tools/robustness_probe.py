"""Developer tool (GPU box): where and how instances of the bench ensemble are lost over a long stretch of the schedule.
usage: python tools/robustness_probe.py [ticks] [batch]   env: SIGMA=<scale of the initial-state noise> DOFS=all|upper ITERS=<n> WALK=1 CLOSED=1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi, ensemble as E
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 300
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
scale = float(os.environ.get("SIGMA", "1"))
pd = FullDynamicsProblem(horizon=100, complete_model=True)
prob = pd.build()
nv = prob.stages[0].xspace.model.nv
dofs = None if os.environ.get("DOFS", "all") == "all" else np.arange(18, nv)  # upper: everything above the two 6-dof legs
x0s = E.ensemble_initial_states(prob.x0_init, prob.stages[0].xspace, batch, sigma_q=0.02 * scale, sigma_v=0.05 * scale, perturb_dofs=dofs)
closed = (10, pd.dt / 10) if os.environ.get("CLOSED") else None
e = E.EnsembleMPC(pd, batch=batch, library=_capi.load_hip_library(), x0=x0s, closed_loop=closed, tick_reuse=True)
e.options.riccati_legs = 4; e.native.set_options(e.options)
e.prepare_schedule(ticks + 4)
st = e.cold_solve(max_iters=100)
print("cold: converged %d/%d" % (sum(bool(s.converged) for s in st), batch))
if os.environ.get("WALK"):
    e.enable_walk(per_instance=bool(os.environ.get("PERINST")))  # PERINST=1: every instance replans from its own measured foot poses
for env, field in (("DYN_SCALE", "dyn_al_scale"), ("ARMIJO", "ls_armijo_c1"), ("REG_INIT", "reg_init"), ("MU_INIT", "mu_init"), ("MU_FACTOR", "bcl_mu_update_factor")):
    if os.environ.get(env):  # the knobs the scripts leave at upstream defaults that this build could not pin
        setattr(e.options, field, float(os.environ[env])); e.native.set_options(e.options)
if os.environ.get("ITERS"):
    e.options.max_iters = int(os.environ["ITERS"]); e.native.set_options(e.options)
extra_total = 0
if os.environ.get("ISOLATE"):  # mpc_set_failure_policy(1) + re-seeding lost instances from the nominal one
    e.enable_failure_isolation(auto_revive=True, source=0)
if os.environ.get("NOSETUP"):  # keep multipliers (and mu) across ticks: no per-tick setup()
    e.native.setup = lambda: None
hist = []
for t in range(1, ticks + 1):
    try:
        st = e.step()
    except RuntimeError as ex:
        print("tick", t, "error:", str(ex)[-90:])
        bad = int(str(ex).split("instance ")[1].split()[0])
        for tt, h in enumerate(hist[-12:]):
            print("  tick %d instance %d: cost %.4e merit %.4e prim %.2e dual %.2e alpha %.3g ls %d mu %.1e" % ((len(hist) - 12 + tt + 1, bad) + h[bad]))
        c = np.array([h[0] for h in hist[-1]])
        print("  costs at the last good tick: nominal %.3e median %.3e, instances above 10x nominal: %s" % (c[0], np.median(c), np.nonzero(c > 10 * abs(c[0]))[0].tolist()))
        break
    extra_total += sum(int(s.num_iters) - 1 for s in st if s.num_iters > 1)
    if os.environ.get("ADAPT"):  # two iterations on the tick after one that ended with a large primal infeasibility somewhere in the ensemble
        want = 2 if max(s.prim_infeas for s in st) > float(os.environ["ADAPT"]) else 1
        if want != e.options.max_iters:
            e.options.max_iters = want; e.native.set_options(e.options)
        adapt_ticks = globals().get("adapt_ticks", 0) + (want == 2)
    hist.append([(s.traj_cost, s.merit, s.prim_infeas, s.dual_infeas, s.alpha, s.ls_steps, s.mu) for s in st])
    if t % 50 == 0:
        c = np.array([h[0] for h in hist[-1]]); al = np.array([h[4] for h in hist[-1]])
        print("tick %4d cost nominal %.3e median %.3e max %.3e | alpha<1: %d | iterations beyond one per tick so far %d (%.2f %% of instance ticks)" % (
            t, c[0], np.median(c), c.max(), int((al < 1).sum()), extra_total, 100.0 * extra_total / (t * batch)))
else:
    if os.environ.get("ADAPT"):
        print("ticks run with two iterations:", globals().get("adapt_ticks", 0))
    print("no failure in", ticks, "ticks" + ("; instances lost and revived: %d (%s ...)" % (e.revived, [tuple(r) for r in e.lost[:6]]) if os.environ.get("ISOLATE") else ""))
