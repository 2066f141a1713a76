#!/bin/bash
# MIOpen's solver search for every convolution shape of the bench workloads, kept as a user find-db (what MIOPEN_USER_DB_PATH points to)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/miopen_db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --workload imagenet_free_at --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/finddb_fa.json 2> gpurun_out/finddb_fa.err
echo "free-AT with search: $(( $(date +%s) - t0 )) s"; grep -o '"value": [0-9.]*' gpurun_out/finddb_fa.json | head -1
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --other-steps 3 > gpurun_out/finddb_default.json 2> gpurun_out/finddb_default.err
ls -la gpurun_out/miopen_db; du -sh gpurun_out/miopen_db
t0=$(date +%s)
timeout -k 10 600 python3 bench.py --workload imagenet_free_at --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/finddb_fa2.json 2> gpurun_out/finddb_fa2.err
echo "free-AT again, db present: $(( $(date +%s) - t0 )) s"; grep -o '"value": [0-9.]*' gpurun_out/finddb_fa2.json | head -1
