#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 ``*_kernel_trace.csv``, with the early-exit launches separated.

rocprofv3's own ``--stats`` average mixes the real launches of a solver kernel with the launches that return
immediately (passes enqueued after every instance has converged, cold-solve tail): this prints both, so the
figure to hold against bench.py's ``roofline.avg_kernel_ms`` is ``avg_real_ms`` (launches >= 10 % of the max).

usage: trace_summary.py KERNEL_TRACE.csv [OUT.csv]
"""
import csv
import sys
from collections import defaultdict


def main():
    rows = defaultdict(list)
    with open(sys.argv[1], newline="") as fh:
        for r in csv.DictReader(fh):
            rows[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    lines = ["kernel,calls,total_ms,avg_ms,real_calls,avg_real_ms,max_ms"]
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        mx = max(v)
        real = [x for x in v if x >= 0.1 * mx]
        lines.append('"%s",%d,%.3f,%.4f,%d,%.4f,%.4f' % (k, len(v), sum(v), sum(v) / len(v), len(real), sum(real) / len(real), mx))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main()
