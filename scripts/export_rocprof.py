#!/usr/bin/env python3
"""rocprofv3 *_kernel_stats.csv -> {bench.py kernel family: average microseconds per launch} (profiles/rocprof_kernel_us.json),
the figure bench.py carries next to its live HIP-event measurement (roofline.rocprofv3_avg_us)."""
import csv
import json
import sys


def _plain_wino(name):
    """the Winograd kernels without a BatchNorm folded in: wino3x3_map4_kernel<0, false, false>, wino3x3_pc_kernel<MAP, 0, false, 0>"""
    return "wino3x3_map4_kernel<0, false, false>" in name or any("wino3x3_pc_kernel<%d, 0, false, 0>" % m in name for m in (4, 8, 16))


def _match(pat, name):
    """every '*'-separated piece of `pat` occurs in `name`, in order (PLAIN / FUSED: the two Winograd families)"""
    if pat.endswith("*PLAIN") or pat.endswith("*FUSED"):
        return pat.split("*")[0] in name and _plain_wino(name) == pat.endswith("PLAIN")
    pos = 0
    for piece in pat.split("*"):
        pos = name.find(piece, pos)
        if pos < 0:
            return False
        pos += len(piece)
    return True


FAMILIES = {  # bench.py family -> substring of the kernel symbol (every template instance of the bench shape)
    "ee_chain_fwd": "chain_fwd_kernel", "ee_chain_bwd": "chain_bwd_kernel", "ee_frontend_fwd": "edge_fwd_kernel",
    "ee_frontend_bwd": "edge_bwd_saved_kernel", "ee_hfs": "hfs_*kernel<0", "ee_hfs_square_fwd": "hfs_*kernel<1", "ee_hfs_square_bwd": "hfs_*kernel<2",
    "ee_pgd_step": "map3_kernel*PgdStepOp", "ee_pgd_step_bcast": "pgd_step_bcast_kernel", "ee_square_draw": "square_draw_kernel",
    "ee_wino3x3": "wino3x3_*PLAIN", "ee_wino3x3_fused": "wino3x3_*FUSED", "ee_conv3x3s2_small_fwd": "conv3s2_fwd_mfma_kernel", "ee_conv3x3s2_small_bwd_data": "conv3s2_bwd_mfma_kernel", "ee_ce": "ce_kernel",
}
rows = list(csv.DictReader(open(sys.argv[1])))
out = {"_provenance": sys.argv[3] if len(sys.argv) > 3 else ""}
for fam, pat in FAMILIES.items():
    tot = calls = 0
    for r in rows:
        if _match(pat, r["Name"]):
            tot += float(r["TotalDurationNs"])
            calls += int(r["Calls"])
    if calls:
        out[fam] = round(tot / calls / 1e3, 3)
        out[fam + "__calls"] = calls
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
