#!/bin/bash
# Register / scratch / occupancy table of every kernel in one csrc/*.hip file (compile-only, no GPU):  scripts/kernel_regs.sh ee_wino.hip
cd "$(dirname "$0")/../edge-enhancement_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -fno-gpu-rdc \
    -I../../include -I. -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("spill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = m.group(1)
print("%-70s %5s %5s %7s %5s %4s %7s" % ("kernel", "vgpr", "agpr", "scratch", "spill", "occ", "lds"))
for r in rows:
    print("%-70s %5s %5s %7s %5s %4s %7s" % (r["name"][:70], r.get("vgpr"), r.get("agpr"), r.get("scratch"), r.get("spill"), r.get("occ"), r.get("lds")))
'
