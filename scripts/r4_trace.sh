#!/bin/bash
# round 4: kernel trace of one bench workload:  scripts/r4_trace.sh <tag> <workload> [steps]
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
tag=$1; wl=$2; steps=${3:-5}
out=$PWD/gpurun_out
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o t -- python3 $out/../bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-other-workloads > $out/${tag}_trace.log 2>&1 || { tail -30 $out/${tag}_trace.log; exit 1; }
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/${tag}_${wl}_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':90s} {'calls':>8s} {'avg_us':>9s} {'total_ms':>9s} {'pct':>6s}")
for r in rows[:40]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>8s} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}")
PY
head -30 $out/${tag}_${wl}_kernel_stats.txt
