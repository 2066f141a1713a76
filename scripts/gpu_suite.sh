#!/bin/bash
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3e_gputests.log 2>&1 || { tail -40 gpurun_out/r3e_gputests.log; exit 1; }
tail -3 gpurun_out/r3e_gputests.log
python3 __graft_entry__.py smoke 2>&1 | tail -2
