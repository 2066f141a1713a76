#!/bin/bash
# rocprofv3 kernel durations of the isolated kernels (scripts/kbench.py).  Usage: scripts/prof_kbench.sh <tag> [kbench args]
set -e
tag=$1; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 scripts/kbench.py "$@" > gpurun_out/kprof_$tag.log 2>&1
f=$(find gpurun_out/kprof_$tag -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_stats.py "$f" 14 | tee gpurun_out/kprof_$tag.summary.txt
