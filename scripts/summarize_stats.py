#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (top kernels by total time)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over %d distinct kernels" % (tot / 1e6, len(rows)))
print("%-9s %-8s %-10s %-6s  %s" % ("total_ms", "calls", "avg_us", "pct", "kernel"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-9.3f %-8s %-10.2f %-6.2f  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3,
                                             100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
