"""GPU diagnostic: consistency of the co-state at the cuts (P_c dx_c + p_c against theta) for HIP and the oracle."""
import numpy as np
from mpc_benchmark_amd import _capi
from tests import _oracle
from tests.test_gpu_legs import _one_iteration

kind, N, legs, complete = "fulldynamic", 16, 4, True
hip, orc = _capi.load_hip_library(), _oracle.load()
res = {}
for tag, lib, L in (("hip", hip, legs), ("orc", orc, legs), ("hip1", hip, 1), ("orc1", orc, 1)):
    pd, s = _one_iteration(lib, kind, N, L, complete)
    res[tag] = s._native
n = pd.space.ndx
starts = [j * N // legs for j in range(legs)] + [N]
for j in range(legs - 1):
    c = starts[j + 1]
    for tag in ("hip", "orc"):
        nat = res[tag]
        P = nat.debug_get("P", c).reshape(n, n); p = nat.debug_get("p", c); dx = nat.debug_get("dx", c)
        th = nat.debug_get("theta", j)
        lam = P @ dx + p
        dP = nat.debug_get("calP", j).reshape(n, n); cp = nat.debug_get("calp", j)
        zc = nat.debug_get("zc", j); Zx = nat.debug_get("Zx", j).reshape(n, n)
        dxs = nat.debug_get("dx", starts[j])
        print("cut %d %s: |P dx + p - theta|/|theta| = %.3e  |theta| %.3e  |dP x + calp - theta| %.3e  |Zx x_j + zc - dx_c|/|dx_c| %.3e |dP|/|P| %.3e" % (
            c, tag, np.max(np.abs(lam - th)) / np.max(np.abs(th)), np.max(np.abs(th)), np.max(np.abs(dP @ dx + cp - th)),
            np.max(np.abs(Zx @ dxs + zc - dx)) / np.max(np.abs(dx)), np.max(np.abs(dP)) / np.max(np.abs(P))))
    P = res["orc1"].debug_get("P", c).reshape(n, n)
    d = res["hip"].debug_get("dx", c) - res["hip1"].debug_get("dx", c)
    d2 = res["orc"].debug_get("dx", c) - res["orc1"].debug_get("dx", c)
    dxr = res["orc1"].debug_get("dx", c)
    print("     P-weighted cut state error: hip legs-serial %.3e  orc legs-serial %.3e  (|P dx| %.3e)" % (np.max(np.abs(P @ d)), np.max(np.abs(P @ d2)), np.max(np.abs(P @ dxr))))
    for name in ("P", "p"):
        a, b = res["hip"].debug_get(name, c), res["orc"].debug_get(name, c)
        print("     %s at cut: hip vs orc (legs) %.3e" % (name, np.max(np.abs(a - b)) / np.max(np.abs(b))))
