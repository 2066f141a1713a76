#!/usr/bin/env python3
"""Where a steady-state step's time goes, from a rocprofv3 kernel trace CSV: per kernel name, the launches, the summed
duration and the summed idle gap that PRECEDES each launch (start - previous end), over the last `tail_ms` of the trace.
Usage: trace_breakdown.py <kernel_trace.csv> [tail_ms=100] [rows=45]"""
import collections
import csv
import sys

tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = max(r[1] for r in rows)
sub = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
agg = collections.defaultdict(lambda: [0, 0, 0])
pred = collections.defaultdict(collections.Counter)  # kernel -> what ran right before its large gaps
prev_end, prev_name = sub[0][0], "-"
for s, e, name in sub:
    a = agg[name]
    a[0] += 1
    a[1] += e - s
    a[2] += max(0, s - prev_end)
    if s - prev_end > 10000:
        pred[name][prev_name[:60]] += 1
    prev_end, prev_name = max(prev_end, e), name
span = prev_end - sub[0][0]
busy = sum(a[1] for a in agg.values())
gap = sum(a[2] for a in agg.values())
print("last %.1f ms: %d launches, busy %.2f ms (%.1f %%), idle before launches %.2f ms" % (span / 1e6, len(sub), busy / 1e6, 100.0 * busy / span, gap / 1e6))
print("%-8s %-9s %-9s %-8s %-8s %s" % ("calls", "kern_ms", "gap_ms", "avg_us", "gap_us", "kernel"))
for name, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:nrows]:
    print("%-8d %-9.3f %-9.3f %-8.2f %-8.2f %s" % (a[0], a[1] / 1e6, a[2] / 1e6, a[1] / a[0] / 1e3, a[2] / a[0] / 1e3, name[:110]))
print("\nlaunches preceded by an idle gap > 10 us, and what ran before them:")
for name, c in sorted(pred.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print("  %-70s %s" % (name[:70], ", ".join("%s x%d" % kv for kv in c.most_common(3))))
