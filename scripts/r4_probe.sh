#!/bin/bash
# round 4: a probe script under rocprofv3 --kernel-trace --stats:  scripts/r4_probe.sh <tag> <script.py>
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
tag=$1; scr=$PWD/$2
out=$PWD/gpurun_out
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python3 $scr > $out/${tag}_probe.log 2> $out/${tag}_probe.err || { tail -30 $out/${tag}_probe.err; exit 1; }
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/${tag}_probe_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':90s} {'calls':>8s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>8s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}")
PY
cat $out/${tag}_probe.log; head -12 $out/${tag}_probe_kernel_stats.txt
