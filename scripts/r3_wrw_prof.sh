#!/bin/bash
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
R=$PWD
cd /tmp
PROBE_OWN_ONLY=1 PROBE_EAGER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_wrw -- python3 $R/scripts/wrw_probe.py 100 > $R/gpurun_out/wrw_prof.log 2>&1
f=$(find /tmp/prof_wrw -name "*kernel_stats.csv" | head -1)
cd $R
python3 scripts/summarize_stats.py "$f" 60 > gpurun_out/wrw_kernel_stats.txt
cut -c1-180 gpurun_out/wrw_kernel_stats.txt
