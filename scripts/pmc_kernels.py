#!/usr/bin/env python3
"""Mean of every collected counter per kernel from a rocprofv3 --pmc run directory.  Usage: pmc_kernels.py <dir> [substring filter]"""
import csv, glob, os, sys
from collections import defaultdict
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(f)):
    if flt in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:90], r["Counter_Name"])
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print("%-92s %-28s %14.2f  (%d)" % (k[0], k[1], acc[k] / cnt[k], cnt[k]))
