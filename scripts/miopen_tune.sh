#!/bin/bash
# Exhaustive MIOpen tuning of the bench workload's convolutions into a user database under gpurun_out/ (copied back).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/miopen_db
( while true; do sleep 50; echo "tick $(date +%s) $(ls gpurun_out/miopen_db | wc -l) files $(du -sk gpurun_out/miopen_db | cut -f1) KB"; done ) &
TICK=$!
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
export MIOPEN_FIND_ENFORCE=${1:-3}
timeout -k 10 ${2:-900} python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
kill $TICK
unset MIOPEN_FIND_ENFORCE
echo "--- with the tuned db, normal find:"
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
ls -la gpurun_out/miopen_db | head
