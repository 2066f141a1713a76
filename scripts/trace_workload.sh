#!/bin/bash
# kernel trace of one bench workload -> steady-state breakdown:  scripts/trace_workload.sh <workload> <tag>   (STEPS=80 for a pure steady-state window)
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
R=$PWD
wl=$1
tag=${2:-r3wl}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --workload $wl --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-other-workloads > $R/gpurun_out/${tag}_bench_profiled.json 2> $R/gpurun_out/${tag}_prof.err
cd $R
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$t" 200 70 > gpurun_out/${tag}_trace_breakdown.txt
head -3 gpurun_out/${tag}_trace_breakdown.txt
