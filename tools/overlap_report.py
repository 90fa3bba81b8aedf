#!/usr/bin/env python3
"""Developer tool: from a rocprofv3 kernel trace (``*kernel_trace.csv``) report how much device time had 1, 2, ... kernels
in flight and, per kernel name, the share of its run time during which a kernel of ANOTHER queue was also running
(do the shards of ``bench.py --streams S`` actually overlap?).   usage: overlap_report.py KERNEL_TRACE.csv [last_fraction]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1], newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")))
    rows.sort()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)  # steady state: the last part of the trace
    rows = [r for r in rows if r[0] >= t_lo]
    ev = []
    for i, (s, e, _, _) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, -1, i))
    ev.sort()
    live = set()
    depth_time = defaultdict(int)
    shared = defaultdict(int)
    total = defaultdict(int)
    last = ev[0][0]
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            depth_time[len(live)] += dt
            queues = defaultdict(int)
            for j in live:
                queues[rows[j][3]] += 1
            for j in live:
                total[rows[j][2]] += dt
                if len(queues) > 1:
                    shared[rows[j][2]] += dt
        last = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    span = sum(depth_time.values())
    print("window %.1f ms, %d dispatches, queues: %s" % (span / 1e6, len(rows), sorted({r[3] for r in rows})))
    for k in sorted(depth_time):
        print("  %d kernels in flight: %6.2f %%" % (k, 100.0 * depth_time[k] / span))
    for k, v in sorted(total.items(), key=lambda kv: -kv[1])[:10]:
        print("  %-40s %8.2f ms busy, %5.1f %% of it with another queue active" % (k[:40], v / 1e6, 100.0 * shared[k] / v))


if __name__ == "__main__":
    main()


def chain_report(path, frac):
    """Per kernel name: mean duration and mean idle time of its queue before it started (steady-state window)."""
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "0")))
    rows.sort()
    t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1.0 - frac)
    last_end = {}
    dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    for s, e, n, q in rows:
        if s >= t_lo and q in last_end:
            dur[n] += e - s; gap[n] += max(0, s - last_end[q]); cnt[n] += 1
        last_end[q] = max(e, last_end.get(q, 0))
    nq = len({r[3] for r in rows if r[0] >= t_lo})
    print("per-queue chain (mean over %d queues): kernel, launches, mean ms, mean queue-idle-before ms" % nq)
    for n in sorted(dur, key=lambda k: -(dur[k] + gap[k])):
        print("  %-40s %5d %8.3f %8.3f" % (n[:40], cnt[n], dur[n] / cnt[n] / 1e6, gap[n] / cnt[n] / 1e6))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "chain":
    chain_report(sys.argv[1], float(sys.argv[2]))
