#!/bin/bash
# HBM traffic of the hand-written kernels from the TCC counters (guides/MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots"):
# FETCH_SIZE and WRITE_SIZE do not fit one pass (3 + 2 of 4 TCC slots) -> two runs, counters only (+ kernel trace).
# Usage: scripts/prof_pmc.sh <tag> [kbench args]
set -e
tag=$1; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$ctr -- python3 scripts/kbench.py --iters 20 "$@" > gpurun_out/pmc_${tag}_$ctr.log 2>&1
done
python3 scripts/summarize_pmc.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE | tee gpurun_out/pmc_$tag.summary.txt
