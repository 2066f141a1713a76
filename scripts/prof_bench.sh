#!/bin/bash
# rocprofv3 kernel trace of bench.py (run on the GPU box through gpurun).  Usage: scripts/prof_bench.sh <tag> [bench args]
set -e
tag=$1; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
python3 scripts/summarize_stats.py "$f" > gpurun_out/prof_$tag.summary.txt
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
head -45 gpurun_out/prof_$tag.summary.txt
