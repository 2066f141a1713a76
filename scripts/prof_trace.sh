#!/bin/bash
# rocprofv3 kernel trace of bench.py reduced to a steady-state breakdown (scripts/trace_breakdown.py); the raw trace is deleted.
# Usage: scripts/prof_trace.sh <tag> <tail_ms> [bench args]
set -e
tag=$1; tail_ms=$2; shift; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python3 bench.py "$@" > gpurun_out/trace_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$f" "$tail_ms" 60 > gpurun_out/trace_$tag.breakdown.txt
tail -1 gpurun_out/trace_$tag.log | cut -c1-300
cat gpurun_out/trace_$tag.breakdown.txt
