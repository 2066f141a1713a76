#!/bin/bash
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
R=$PWD
cd /tmp
for b in 13 50 100 200; do
PROBE_OWN_ONLY=1 PROBE_EAGER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wrw_$b -- python3 $R/scripts/wrw_probe.py $b > $R/gpurun_out/wrw_prof_$b.log 2>&1
f=$(find /tmp/prof_wrw_$b -name "*kernel_stats.csv" | head -1)
echo "B=$b"; python3 $R/scripts/summarize_stats.py "$f" 60 | grep -E "wrw" | cut -c1-110
done
