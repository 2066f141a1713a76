#!/usr/bin/env python3
"""Idle gaps of the steady state from a rocprofv3 kernel trace CSV: every gap above `min_us` in the last `tail_ms`, with the kernel
before and after it, and the total.  Usage: trace_gaps.py <kernel_trace.csv> [tail_ms=150] [min_us=15]"""
import csv
import sys

tail_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t_end = rows[-1][1]
sub = [r for r in rows if r[0] >= t_end - tail_ms * 1e6]
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
prev_end, prev = sub[0][1], sub[0][2]
tot = big = 0.0
t0 = sub[0][0]
for s, e, n in sub[1:]:
    gap = (s - prev_end) / 1e3
    if gap > 0:
        tot += gap
    if gap > min_us:
        big += gap
        print("t=%9.1f us  gap %7.1f us  after %-60s before %s" % ((s - t0) / 1e3, gap, short(prev), short(n)))
    if e > prev_end:
        prev_end, prev = e, n
print("window %.1f ms: idle %.1f us in all gaps, %.1f us in gaps > %.0f us" % ((prev_end - t0) / 1e6, tot, big, min_us))
