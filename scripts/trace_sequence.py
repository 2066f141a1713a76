#!/usr/bin/env python3
"""The kernel sequence of ONE steady-state PGD iteration (from one chain_fwd launch to the next) and of one parameter update,
from a rocprofv3 kernel trace CSV: name, duration, idle gap before it.  Usage: trace_sequence.py <kernel_trace.csv> [marker substring]"""
import csv
import sys

marker = sys.argv[2] if len(sys.argv) > 2 else "chain_fwd_kernel"
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
# the last complete attack: 10 consecutive marker launches close together; print the interval between the 5th and 6th from the end
a, b = marks[-6], marks[-5]


def show(lo, hi, title):
    print("%s: %d launches, %.1f us wall" % (title, hi - lo, (rows[hi][0] - rows[lo][0]) / 1e3))
    prev = rows[lo - 1][1]
    tot = 0
    for s, e, n in rows[lo:hi]:
        n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
        print("  %7.2f us  gap %6.2f  %s" % ((e - s) / 1e3, max(0, s - prev) / 1e3, n[:120]))
        prev = max(prev, e)
        tot += e - s
    print("  busy %.1f us" % (tot / 1e3))


show(a, b, "one PGD iteration")
# the update = what follows the last marker of an attack until the next attack's first marker: find the largest gap between consecutive markers
gaps = [(marks[i + 1] - marks[i], i) for i in range(len(marks) - 12, len(marks) - 1)]
_, i = max(gaps)
show(marks[i], marks[i + 1], "last iteration of an attack + the parameter update + the next attack's first launches")
