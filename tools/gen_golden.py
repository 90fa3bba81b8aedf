#!/usr/bin/env python3
"""Golden vectors from the REAL reference stack.  Runs only where ``import aligator, pinocchio`` succeeds —
Aligator >= 0.10 / Pinocchio >= 2.9.1 as the reference's README pins them; neither is installable in the build container or on
the GPU box, so this script has NOT been executed there: it is the route by which someone with the stack moves the repo's parity
from "HIP == the repo's own oracle" to "HIP == Aligator".

What it does
  1. exports ``talos_synth_v1`` (mpc_benchmark_amd/robot/talos_synth.py: the committed joint table) as a real ``pinocchio.Model``
     (complete nq = 39 and the locked-joint reduction nq = 29 of talos_utils.py:31-41);
  2. builds the three OCPs with the REAL modules through this repo's own builders (mpc_benchmark_amd/problems/*.py are written
     against the ``aligator`` API: the module globals are re-bound to the real packages);
  3. dumps, per problem, to tests/golden/aligator_<problem>.npz:
       model_*           mass, CoM and sole placements of the exported model at the reference posture (export check)
       eval_<kind>_*     for one stage of every contact kind at a seeded (x, u): stage.evaluate / computeFirstOrderDerivatives /
                         computeSecondOrderDerivatives outputs — cost value, Lx, Lu, Lxx, Lxu, Luu, xnext, dynamics Jx, Ju,
                         every constraint's value, Jx, Ju (the 17-row wrench-cone matrix is the cone's Ju / d lambda: also dumped
                         as ``cone_A`` from the residual when exposed)
       iter1_*, conv_*   results.xs / us / controlFeedbacks()[0] (+ iteration counts, criteria) after max_iters = 1 and after the cold
                         solve of the scripts (max_iters = 100, TOL 1e-5: fulldynamic_talos.py:374-397, kinodynamic_talos.py:281-304,
                         centroidal_talos.py:265-288)
  4. tests/test_golden_aligator.py consumes the files (auto-skips while they are absent).

Rehearsal: ``generate(out_dir, ..., standin_lib=<library handle>)`` runs steps 2 - 4 against this repo's own mirror (tests/test_golden_aligator.py::
test_rehearsal_*, into a temporary directory): every line of the dumps and of the six consumers has executed before someone with the stack runs them.  What
it writes is NOT golden — the numbers are this build's own.

usage:  python tools/gen_golden.py [--problems fulldynamic,kinodynamic,centroidal] [--horizon 20] [--out tests/golden]
"""
import argparse
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def real_stack():
    try:
        import aligator
        import pinocchio
    except Exception as e:  # noqa: BLE001
        print("gen_golden: the reference stack is not importable here (%s): nothing to do." % e)
        return None, None
    if "mpc_benchmark_amd" in getattr(aligator, "__file__", ""):
        print("gen_golden: `aligator` resolves to this repo's mirror, not to the real package: nothing to do.")
        return None, None
    return aligator, pinocchio


def standin_stack():
    """``--standins`` / MPC_ALIGATOR_STANDINS=1: this repo's own mirror and its Pinocchio stand-in in the place of the real packages — a PLUMBING REHEARSAL
    of steps 2 - 4 (builders, dumps, file layout, consumers), so that the first person with the real stack is not the first to execute this code.  The
    numbers it writes come from this build itself: they are NOT golden vectors and prove no parity."""
    from mpc_benchmark_amd import aligator as mirror
    from mpc_benchmark_amd.robot import minipin
    return mirror, minipin


def export_models(pin):
    """talos_synth_v1 as real pinocchio models: (complete, reduced, q_complete, q_reduced) like loadTalos() (talos_utils.py:31-41)."""
    from mpc_benchmark_amd.robot import talos_synth as ts
    kinds = {"JointModelRX": pin.JointModelRX, "JointModelRY": pin.JointModelRY, "JointModelRZ": pin.JointModelRZ}
    m = pin.Model()
    m.name = "talos_synth_v1"
    big = 1e3
    root = m.addJoint(0, pin.JointModelFreeFlyer(), pin.SE3.Identity(), "root_joint")
    m.appendBodyToJoint(root, pin.Inertia(ts._BASE_MASS, np.array(ts._BASE_COM), ts._box(ts._BASE_MASS, *ts._BASE_BOX)), pin.SE3.Identity())
    m.addFrame(pin.Frame("base_link", root, 0, pin.SE3.Identity(), pin.FrameType.BODY))
    for name, parent, kind, trans, mass, com, dims, effort, (lo, hi) in ts._TREE:
        pid = m.getJointId(parent)
        jid = m.addJoint(pid, kinds[kind](), pin.SE3(np.eye(3), np.array(trans, dtype=float)), name,
                         np.array([effort]), np.array([big]), np.array([lo]), np.array([hi]))
        m.appendBodyToJoint(jid, pin.Inertia(mass, np.array(com, dtype=float), ts._box(mass, *dims)), pin.SE3.Identity())
        m.addFrame(pin.Frame(name.replace("_joint", "_link"), jid, 0, pin.SE3.Identity(), pin.FrameType.BODY))
    for side in ("left", "right"):
        jid = m.getJointId("leg_%s_6_joint" % side)
        m.addFrame(pin.Frame("%s_sole_link" % side, jid, 0, pin.SE3(np.eye(3), np.array([0.0, 0.0, -0.107])), pin.FrameType.OP_FRAME))
    # free-flyer limits (the scripts only read the joint part [7:] / [6:])
    q = pin.neutral(m)
    q[2] = ts.BASE_HEIGHT
    q[7:] = ts._HALF_SITTING_JOINTS
    m.referenceConfigurations["half_sitting"] = q
    reduced = pin.buildReducedModel(m, list(ts.LOCKED_JOINT_IDS), q)
    q_red = np.concatenate([q[:7], [q[7 + j - 2] for j in range(2, m.njoints) if j not in ts.LOCKED_JOINT_IDS]])
    reduced.referenceConfigurations["half_sitting"] = q_red
    return m, reduced, q, q_red


def bind_real_modules(aligator, pin, models):
    """Re-bind the module globals of the problem builders to the real packages and hand them the exported models."""
    from mpc_benchmark_amd.problems import centroidal, common, fulldynamic, kinodynamic
    fake_ts = types.SimpleNamespace(load_talos=lambda: models)
    common.pin = pin
    common.talos_synth = fake_ts
    for mod in (fulldynamic, kinodynamic, centroidal):
        mod.aligator = aligator
        for sub in ("constraints", "dynamics", "manifolds"):
            if hasattr(mod, sub):
                setattr(mod, sub, getattr(aligator, sub))
        if hasattr(mod, "pin"):
            mod.pin = pin
    return {"fulldynamic": fulldynamic.FullDynamicsProblem, "kinodynamic": kinodynamic.KinodynamicProblem, "centroidal": centroidal.CentroidalProblem}


def dump_stage(out, tag, stage, x, u):
    """evaluate + first / second-order derivatives of one StageModel at (x, u); y = the stage's own prediction (zero dynamics gap)."""
    data = stage.createData()
    stage.evaluate(x, u, x, data)
    xnext = np.array(data.dynamics_data.xnext) if hasattr(data.dynamics_data, "xnext") else None
    y = xnext if xnext is not None else x
    stage.evaluate(x, u, y, data)
    stage.computeFirstOrderDerivatives(x, u, y, data)
    stage.computeSecondOrderDerivatives(x, u, y, data)
    cd, dd = data.cost_data, data.dynamics_data
    out[tag + "_x"], out[tag + "_u"] = np.array(x), np.array(u)
    out[tag + "_cost"] = np.array([cd.value])
    for name in ("Lx", "Lu", "Lxx", "Lxu", "Luu"):
        out[tag + "_" + name] = np.array(getattr(cd, name))
    if xnext is not None:
        out[tag + "_xnext"] = xnext
    for name in ("Jx", "Ju", "value"):
        if hasattr(dd, name):
            out[tag + "_dyn_" + name] = np.array(getattr(dd, name))
    for i, cdat in enumerate(data.constraint_data):
        out["%s_c%d_value" % (tag, i)] = np.array(cdat.value)
        out["%s_c%d_Jx" % (tag, i)] = np.array(cdat.Jx)
        out["%s_c%d_Ju" % (tag, i)] = np.array(cdat.Ju)
    cont = getattr(dd, "continuous_data", None)
    if cont is not None and hasattr(cont, "xdot"):
        out[tag + "_xdot"] = np.array(cont.xdot)
        try:
            out[tag + "_wrenches"] = np.concatenate([np.array(c.contact_force.vector) for c in cont.constraint_datas])
        except Exception:  # noqa: BLE001
            pass


def dump_stage_standin(out, tag, stage, x, u, pd, aligator, lib):
    """The rehearsal's stand-in for ``dump_stage``: the mirror's StageData is a placeholder (the native library owns the workspace), so the same arrays
    come from a one-knot problem solved for one iteration by ``lib`` (the caller's library handle: tests pass the CPU checker) — read back through
    ``mpc_debug_get`` exactly as tests/test_golden_aligator.py reads the quantities it compares them with."""
    prob = aligator.TrajOptProblem(x, [stage], aligator.CostStack(stage.xspace, stage.nu))
    solver = pd.make_solver(_native_library=lib)
    solver.linear_solver_choice = aligator.LQ_SOLVER_SERIAL
    solver.max_iters = 1
    solver.corrector_prim_tol = 0.0
    solver.setup(prob)
    solver.run(prob, [x, x], [u])
    nat = solver._native
    try:
        xnext = nat.debug_get("xnext", 0).ravel()
    except RuntimeError:
        xnext = None
    if xnext is not None and xnext.size == np.asarray(x).size:
        solver.run(prob, [x, xnext], [u])  # zero dynamics gap, as dump_stage
    else:
        xnext = None
    n, nu = stage.xspace.ndx, stage.nu
    out[tag + "_x"], out[tag + "_u"] = np.array(x), np.array(u)
    out[tag + "_cost"] = np.array([nat.debug_get("cost", 0)[0]])
    grad = nat.debug_get("grad", 0).ravel()
    H = nat.debug_get("H", 0).reshape(n + nu, n + nu)
    out[tag + "_Lx"], out[tag + "_Lu"] = grad[:n], grad[n:]
    out[tag + "_Lxx"], out[tag + "_Lxu"], out[tag + "_Luu"] = H[:n, :n], H[:n, n:], H[n:, n:]
    if xnext is not None:
        AB = nat.debug_get("AB", 0).reshape(n, n + nu)
        out[tag + "_xnext"], out[tag + "_dyn_Jx"], out[tag + "_dyn_Ju"] = xnext, AB[:, :n], AB[:, n:]
    dims = [int(f.nr) for f in stage.constraints.funcs]
    if dims:
        cval = nat.debug_get("cval", 0).ravel()
        CD = nat.debug_get("CD", 0).reshape(sum(dims), n + nu)
        r0 = 0
        for i, d in enumerate(dims):
            out["%s_c%d_value" % (tag, i)] = cval[r0:r0 + d]
            out["%s_c%d_Jx" % (tag, i)], out["%s_c%d_Ju" % (tag, i)] = CD[r0:r0 + d, :n], CD[r0:r0 + d, n:]
            r0 += d


def dump_problem(out, aligator, name, builder, horizon, standin_lib=None):
    pd = builder(horizon=horizon)
    rb = getattr(pd, "robot", None)
    if rb is not None:
        out["model_mass"] = np.array([rb.mass])
        out["model_com0"] = np.array(rb.com0)
        out["model_q0"] = np.array(rb.q0)
        for i, M in enumerate(rb.foot_placements):
            out["model_sole%d_R" % i], out["model_sole%d_p" % i] = np.array(M.rotation), np.array(M.translation)
    rng = np.random.default_rng(20250304)
    # one stage per contact kind at a seeded point near the reference posture
    kinds = {"double": [True, True], "left": [True, False], "right": [False, True]}
    for kname, cs in kinds.items():
        try:
            if name == "fulldynamic":
                lf, rf = rb.foot_placements
                st = pd.create_stage(cs, lf.copy(), rf.copy())
            elif name == "kinodynamic":
                lf, rf = rb.foot_placements
                st = pd.create_stage(cs, lf.copy(), rf.copy(), pd.urefs[0])
            else:
                if cs not in pd.contact_phases:
                    continue
                st = pd.stage_for_tick(pd.contact_phases.index(cs))
            space = st.xspace
            x = space.integrate(np.array(pd.x0), 0.03 * rng.standard_normal(space.ndx))
            u = (getattr(pd, "u_init", np.zeros(st.nu)) + rng.standard_normal(st.nu) * (15.0 if name == "fulldynamic" else 1.0))
            if standin_lib is not None:
                dump_stage_standin(out, "eval_" + kname, st, x, u, pd, aligator, standin_lib)
            else:
                dump_stage(out, "eval_" + kname, st, x, u)
        except Exception as e:  # noqa: BLE001
            print("gen_golden[%s]: stage kind %s skipped: %s" % (name, kname, e))
    # the scripts' solves: one iteration, then to convergence
    for tag, iters in (("iter1", 1), ("conv", 100)):
        prob = pd.build(with_terminal_constraint=True) if name == "fulldynamic" else pd.build()
        solver = pd.make_solver() if standin_lib is None else pd.make_solver(_native_library=standin_lib)
        solver.max_iters = iters
        solver.setup(prob)
        xs, us = pd.initial_guess()
        solver.run(prob, xs, us)
        r = solver.results
        out[tag + "_xs"], out[tag + "_us"] = np.array(r.xs.tolist()), np.array(r.us.tolist())
        out[tag + "_K0"] = np.array(r.controlFeedbacks()[0])
        out[tag + "_stats"] = np.array([r.num_iters, float(r.conv), r.traj_cost, r.prim_infeas, r.dual_infeas])
    out["horizon"] = np.array([horizon])
    out["versions"] = np.array([getattr(aligator, "__version__", "?") if standin_lib is None else "STAND-INS (rehearsal: not golden vectors)"])


def generate(out_dir, problems, horizon, complete=False, standin_lib=None):
    """Steps 1 - 3 for the named problems into ``out_dir`` -> the files written.  ``standin_lib``: the rehearsal (see standin_stack): steps 2 - 3 against the mirror
    with the caller's library handle ; step 1 (the export through the real pinocchio) has no stand-in — the builders load the committed joint table themselves."""
    if standin_lib is None:
        aligator, pin = real_stack()
        if aligator is None:
            return []
        builders = bind_real_modules(aligator, pin, export_models(pin))
    else:
        aligator, _ = standin_stack()
        from mpc_benchmark_amd.problems import centroidal, fulldynamic, kinodynamic
        builders = {"fulldynamic": fulldynamic.FullDynamicsProblem, "kinodynamic": kinodynamic.KinodynamicProblem, "centroidal": centroidal.CentroidalProblem}
    os.makedirs(out_dir, exist_ok=True)
    written = []
    for name in problems:
        out = {}
        b = builders[name]
        dump_problem(out, aligator, name, (lambda horizon, b=b: b(horizon=horizon, complete_model=complete)) if name != "centroidal" else b, horizon, standin_lib=standin_lib)
        path = os.path.join(out_dir, "aligator_%s%s.npz" % (name, "_complete" if complete else ""))
        np.savez_compressed(path, **out)
        print("gen_golden: wrote %s (%d arrays)%s" % (path, len(out), "" if standin_lib is None else "  [STAND-INS: a rehearsal of the plumbing, NOT golden vectors]"))
        written.append(path)
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--problems", default="fulldynamic,kinodynamic,centroidal")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--complete", action="store_true", help="complete model (nq = 39) instead of the scripts' reduced one")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    generate(args.out, args.problems.split(","), args.horizon, complete=args.complete)
    return 0


if __name__ == "__main__":
    sys.exit(main())
