#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into per-launch HBM traffic of the solver kernels.

usage: summarize.py OUT.json FETCH_DIR WRITE_DIR CALIB_FETCH_DIR CALIB_WRITE_DIR CALIB_BYTES

Each *_DIR holds the ``*counter_collection.csv`` of ONE pass (FETCH_SIZE and WRITE_SIZE cannot share a pass on
gfx950, MI355X_MICROARCH.md §rocprofv3 PMC slots).  The calibration passes ran tools/pmc/pmc_calib (a streaming
8 B/lane copy of CALIB_BYTES in and CALIB_BYTES out per launch): bytes-per-counter-unit is derived from it, which
subsumes the guide's "double FETCH_SIZE" correction for this access width.  Early-exit launches of a kernel
(passes enqueued after convergence return immediately) and the launches of the cold-solve tail (most instances
already finished) are dropped: only launches whose counter is >= 90 % of that kernel's maximum count.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_pass(d, counter):
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                rows[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    return rows


def mean_real(v):
    if not v:
        return None, 0
    mx = max(v)
    keep = [x for x in v if x >= 0.9 * mx]  # the full-ensemble launches (the cold-solve tail runs with most instances finished)
    return sum(keep) / len(keep), len(keep)


def main():
    out, fdir, wdir, cfdir, cwdir, cbytes = sys.argv[1:7]
    cbytes = float(cbytes)
    cal_f, _ = mean_real(read_pass(cfdir, "FETCH_SIZE").get("k_calib_copy8", []))
    cal_w, _ = mean_real(read_pass(cwdir, "WRITE_SIZE").get("k_calib_copy8", []))
    res = {"calibration": {"bytes_per_launch_each_way": cbytes, "FETCH_SIZE_per_launch": cal_f, "WRITE_SIZE_per_launch": cal_w,
                           "bytes_per_FETCH_SIZE_unit": cbytes / cal_f if cal_f else None,
                           "bytes_per_WRITE_SIZE_unit": cbytes / cal_w if cal_w else None},
           "kernels": {}}
    fr, wr = read_pass(fdir, "FETCH_SIZE"), read_pass(wdir, "WRITE_SIZE")
    for k in sorted(set(fr) | set(wr)):
        f, nf = mean_real(fr.get(k, []))
        w, nw = mean_real(wr.get(k, []))
        e = {"launches_counted": max(nf, nw), "FETCH_SIZE": f, "WRITE_SIZE": w}
        if f is not None and cal_f:
            e["read_bytes"] = f * cbytes / cal_f
        if w is not None and cal_w:
            e["write_bytes"] = w * cbytes / cal_w
        if "read_bytes" in e and "write_bytes" in e:
            e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
        res["kernels"][k] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
