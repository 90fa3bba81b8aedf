"""The LDS plan of the Riccati sweep (mpc_benchmark_amd/csrc/riccati_layout.h: which operand of a knot lives where in the 160 KB of a workgroup) checked
on the host: the header is plain constexpr C++, compiled here with g++ into a dump of every region of every plan for the problem dimensions the library
instantiates.  Regions that are live at the same time must not overlap:

  * phase 1 (series): Pt | LP | LI ;  phase 2 (products with [A B]): Pt | the rows of [A B] the sweep reads | G_u ;  phase 3 (stage KKT system and value
    update): Lr | LIr | W | ST | CT | VX | Y | SC | LIs — each phase inside [R1, vec), the small vectors behind them;
  * with the overlap layout (ovl: one wavefront factorises Ruu while the others still multiply with [A B]) Lr / LIr must also keep clear of the live rows
    of [A B];
  * plan 3 (round 5) keeps only the rows ks = (n / 2) & ~3 .. np of [A B]; the headline problem (n = 76, m = 32) stays on plan 1, byte for byte what
    round 4 measured."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plans(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ric") / "ric_layout_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "_native", "ric_layout_dump.cpp")])
    out = []
    for line in subprocess.check_output([exe], text=True).splitlines():
        head, regs = line.split(" | ")
        p = {k: int(v) for k, v in (kv.split("=") for kv in head.split())}
        p["regions"] = {name: (int(a), int(b)) for name, a, b in (r.split(":") for r in regs.split())}
        out.append(p)
    return out


def _overlap(a, b):
    return a[1] > 0 and b[1] > 0 and a[0] < b[0] + b[1] and b[0] < a[0] + a[1]


def test_regions_that_live_together_do_not_overlap(plans):
    assert len(plans) == 8 * 4 * 2
    checked = 0
    for p in plans:
        r = p["regions"]
        lo, hi = r["R1"][0], r["vec"][0]
        tag = "n=%d m=%d plan=%d st=%d" % (p["n"], p["m"], p["plan"], p["st"])
        assert r["PT"][0] == 0 and r["PT"][1] <= lo, tag
        phases = {"1": ["LP", "LI"], "2": ["ABlive", "GP"], "3": ["Lr", "LIr", "W", "ST", "CT", "VX", "Y", "SC", "LIs"]}
        for ph, names in phases.items():
            for i, a in enumerate(names):
                assert r[a][1] == 0 or (lo <= r[a][0] and r[a][0] + r[a][1] <= hi), "%s: %s of phase %s leaves [R1, vec)" % (tag, a, ph)
                for b in names[i + 1:]:
                    assert not _overlap(r[a], r[b]), "%s: %s and %s overlap (phase %s)" % (tag, a, b, ph)
        if p["ovl"]:
            for a in ("Lr", "LIr"):
                assert not _overlap(r[a], r["ABlive"]), "%s: %s lies over rows of [A B] that are still read while it is factorised" % (tag, a)
        assert r["vec"][0] + r["vec"][1] <= p["iwork"] and p["total_bytes"] == 8 * p["iwork"] + 4 * (p["c"] + 72), tag
        checked += 1
    assert checked == len(plans)


def test_plan_3_keeps_only_the_rows_the_structured_sweep_reads(plans):
    by = {(p["n"], p["m"], p["plan"], p["st"]): p for p in plans}
    for n, m in ((76, 44), (56, 34), (56, 22), (76, 32)):
        p1, p3 = by[(n, m, 1, 1)], by[(n, m, 3, 1)]
        ks = (n // 2) & ~3
        assert p3["skip"] == ks and p3["regions"]["ABlive"][1] == (p3["np"] - ks) * p3["nzp"] and p1["regions"]["ABlive"][1] == p1["np"] * p1["nzp"]
        assert p3["ovl"] == 1                      # room for the overlapped factorisation of Ruu
        assert p3["total_bytes"] <= p1["total_bytes"]
    # BASELINE config 4 (complete model, kinodynamic: n = 76, m = 44): the plan the library picks (3, Sh^T out of LDS) fits the 160 KB; plan 1 does not
    assert by[(76, 44, 3, 0)]["total_bytes"] <= 160 * 1024 < by[(76, 44, 1, 0)]["total_bytes"]
    assert by[(76, 44, 3, 0)]["regions"]["GP"][1] == 80 * 48   # G_u whole, on chip


def test_headline_plan_is_the_one_round_4_measured(plans):
    p = next(q for q in plans if (q["n"], q["m"], q["plan"], q["st"]) == (76, 32, 1, 1))
    r = p["regions"]
    assert p["ovl"] == 1 and p["skip"] == 0
    assert (r["PT"][0], r["R1"][0], r["ABlive"][0], r["GP"][0], r["W"][0], r["ST"][0], r["Lr"][0], r["LIr"][0], r["vec"][0]) == (0, 6480, 6480, 15440, 6480, 9584, 16400, 17456, 18000)
