"""Numerics probe (CPU, numpy): a TREE over the cuts instead of the chain of k_leg_consensus — legs composed pairwise (the star
product of their condensed forms), cut states by a down-sweep — against the oracle's chain consensus on the same leg records."""
import os, sys
os.environ["MPC_LEGS_PLAIN"] = "1"
os.environ["MPC_LEGS_CHAIN"] = "1"  # the oracle resolves the cuts by its chain: the reference for the numpy tree below
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests import _oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
J = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _oracle.load()
fp = FullDynamicsProblem(horizon=N, complete_model=False)
prob = fp.build(with_terminal_constraint=True)
solver = fp.make_solver(_native_library=lib)
solver.setNumThreads(J)
solver.max_iters = 1
solver.setup(prob)
rng = np.random.default_rng(2)
xs = [fp.space.integrate(fp.x0, 0.02 * rng.standard_normal(fp.space.ndx)) for _ in range(N + 1)]
us = [10.0 * rng.standard_normal(fp.nu) for _ in range(N)]
prob.x0_init = xs[0]
solver.run(prob, xs, us)
nat = solver._native
n = fp.space.ndx
start = [j * N // J for j in range(J)]
get = lambda nm, k, shape=None: (np.array(nat.debug_get(nm, k)).reshape(shape) if shape else np.array(nat.debug_get(nm, k)))
legs = []
for j in range(J):
    s = start[j]
    d = dict(P=get("P", s, (n, n)), p=get("p0", s) if j + 1 < J else get("p", s))
    if j + 1 < J:
        d.update(Lm=get("Lm", s, (n, n)), Sg=get("Sg", s, (n, n)), sg=get("sg", s))
    else:
        d.update(Lm=np.zeros((n, n)), Sg=np.zeros((n, n)), sg=np.zeros(n))
    legs.append(d)
x_chain = [get("dx", start[j]) for j in range(J)]          # state at the start of every leg
th_chain = [get("theta", j) for j in range(J - 1)]          # co-state parameter at the end of leg j


def compose(a, b):
    """node a followed by node b (Pg = 0 at the cut between them): condensed form of the pair + what the down-sweep needs"""
    D = b["P"]
    W = np.linalg.inv(np.eye(n) - a["Sg"] @ D)
    T1, T2, t3 = W @ a["Lm"].T, W @ a["Sg"], W @ (a["Sg"] @ b["p"] + a["sg"])
    ab = dict(P=a["P"] + a["Lm"] @ D @ T1, p=a["p"] + a["Lm"] @ (D @ t3 + b["p"]), Lm=a["Lm"] @ (np.eye(n) + D @ T2) @ b["Lm"],
              Sg=b["Sg"] + b["Lm"].T @ T2 @ b["Lm"], sg=b["sg"] + b["Lm"].T @ t3,
              link=dict(Zx=T1, Zt=T2 @ b["Lm"], zc=t3, D=D, Lb=b["Lm"], pb=b["p"]), left=a, right=b)
    ab["P"] = 0.5 * (ab["P"] + ab["P"].T)
    return ab


level = [dict(l, lo=j, hi=j) for j, l in enumerate(legs)]
depth = 0
while len(level) > 1:
    nxt = []
    for i in range(0, len(level) - 1, 2):
        ab = compose(level[i], level[i + 1]); ab["lo"], ab["hi"] = level[i]["lo"], level[i + 1]["hi"]
        nxt.append(ab)
    if len(level) % 2:
        nxt.append(level[-1])
    level = nxt; depth += 1
root = level[0]
x_tree, th_tree = {0: x_chain[0]}, {}


def down(node, x_in, th_out):
    if "link" not in node:
        return
    lk = node["link"]
    x_mid = lk["Zx"] @ x_in + lk["Zt"] @ th_out + lk["zc"]
    th_mid = lk["D"] @ x_mid + lk["Lb"] @ th_out + lk["pb"]
    cut = node["right"]["lo"]
    x_tree[cut] = x_mid; th_tree[cut - 1] = th_mid
    down(node["left"], x_in, th_mid)
    down(node["right"], x_mid, th_out)


down(root, x_chain[0], np.zeros(n))
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))
print("N=%d legs=%d n=%d tree depth %d" % (N, J, n, depth))
Ps = get("P", 0, (n, n))
print("cut states  tree vs chain: max rel %.3e" % max(rel(x_tree[j], x_chain[j]) for j in range(1, J)))
print("co-states   tree vs chain: max rel %.3e" % max(rel(th_tree[j], th_chain[j]) for j in range(J - 1)))
for j in range(1, J):
    print("   cut %d: |x| %.3e  err %.3e   |theta| %.3e err %.3e" % (j, np.max(np.abs(x_chain[j])), np.max(np.abs(x_tree[j] - x_chain[j])), np.max(np.abs(th_chain[j - 1])), np.max(np.abs(th_tree[j - 1] - th_chain[j - 1]))))
