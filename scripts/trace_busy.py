#!/usr/bin/env python3
"""GPU busy fraction and gap statistics from a rocprofv3 kernel trace CSV (last `frac` of the run = steady state)."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sub = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(sub[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sub)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sub)
gaps = [int(sub[i + 1]["Start_Timestamp"]) - int(sub[i]["End_Timestamp"]) for i in range(len(sub) - 1)]
pos = [g for g in gaps if g > 0]
print("kernels %d  span %.1f ms  busy %.1f ms (%.1f %%)  mean kernel %.1f us" % (len(sub), (t1 - t0) / 1e6, busy / 1e6, 100 * busy / (t1 - t0),
                                                                         busy / len(sub) / 1e3))
print("gaps: median %.1f us  mean %.1f us  total %.1f ms;  > 20 us: %d totalling %.1f ms" % (
    statistics.median(pos) / 1e3, statistics.mean(pos) / 1e3, sum(pos) / 1e6, sum(g > 20000 for g in pos), sum(g for g in pos if g > 20000) / 1e6))
