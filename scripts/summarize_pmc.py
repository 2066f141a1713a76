#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).
Units / corrections per guides/MI355X_MICROARCH.md: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read (128-B requests tallied as 64 B), so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == name:
            k = r["Kernel_Name"]
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}, cnt


fetch, n = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
print("%-12s %-12s %-12s %-8s %s" % ("fetch_MB*2", "write_MB", "traffic_MB", "launches", "kernel"))
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if "anonymous namespace" not in k:
        continue
    fb, wb = 2.0 * fetch[k] * 1024, write.get(k, 0.0) * 1024
    print("%-12.3f %-12.3f %-12.3f %-8d %s" % (fb / 1e6, wb / 1e6, (fb + wb) / 1e6, n[k], k[:110]))
