#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> mean HBM bytes per launch of every bench.py kernel family, merged into
profiles/pmc_traffic.json under the shape key.  Units / corrections per guides/MI355X_MICROARCH.md: counters in KiB; FETCH_SIZE
doubled on gfx950 (128-B read requests are tallied as 64 B); WRITE_SIZE exact.
Usage: pmc_to_json.py <fetch_dir> <write_dir> <shape key, e.g. 100x3x64x64> <json path> [provenance]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def _match(pat, name):
    """every '*'-separated piece of `pat` occurs in `name`, in order"""
    pos = 0
    for piece in pat.split("*"):
        pos = name.find(piece, pos)
        if pos < 0:
            return False
        pos += len(piece)
    return True


FAMILIES = {
    "ee_chain_fwd": "chain_fwd_kernel", "ee_chain_bwd": "chain_bwd_kernel", "ee_frontend_fwd": "edge_fwd_kernel",
    "ee_frontend_bwd": "edge_bwd_saved_kernel", "ee_hfs": "hfs_*kernel<0", "ee_hfs_square_fwd": "hfs_*kernel<1", "ee_hfs_square_bwd": "hfs_*kernel<2",
    "ee_pgd_step": "map3_kernel*PgdStepOp", "ee_pgd_step_bcast": "pgd_step_bcast_kernel", "ee_square_draw": "square_draw_kernel",
    "ee_wino3x3": "wino3x3_", "ee_conv3x3s2_small_fwd": "conv3s2_fwd_mfma_kernel", "ee_conv3x3s2_small_bwd_data": "conv3s2_bwd_mfma_kernel",
}


def load(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        raise SystemExit("no counter_collection.csv under " + d)
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") == name:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return acc, cnt


fa, fc = load(sys.argv[1], "FETCH_SIZE")
wa, wc = load(sys.argv[2], "WRITE_SIZE")
shape, path = sys.argv[3], sys.argv[4]
try:
    out = json.load(open(path))
except (OSError, ValueError):
    out = {}
entry = out.setdefault(shape, {})
for fam, pat in FAMILIES.items():
    fb = sum(v for k, v in fa.items() if _match(pat, k))
    fn = sum(v for k, v in fc.items() if _match(pat, k))
    wb = sum(v for k, v in wa.items() if _match(pat, k))
    wn = sum(v for k, v in wc.items() if _match(pat, k))
    if fn and wn:
        entry[fam] = int(round(2.0 * fb * 1024 / fn + wb * 1024 / wn))
        print("%-20s fetch*2 %.3f MB  write %.3f MB  -> %.3f MB per launch (%d launches)" % (fam, 2.0 * fb * 1024 / fn / 1e6, wb * 1024 / wn / 1e6, entry[fam] / 1e6, fn))
if len(sys.argv) > 5:
    out["_provenance_" + shape] = sys.argv[5]
json.dump(out, open(path, "w"), indent=1)
