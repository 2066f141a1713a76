#!/bin/bash
# round 4: a subset of the GPU tests + short bench legs of the named workloads:  scripts/r4_quick.sh <tag> "<pytest args>" <workload> ...
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
tag=$1; shift
sel=$1; shift
if [ -n "$sel" ]; then
  timeout -k 10 900 python3 -m pytest $sel -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1 || { tail -60 gpurun_out/${tag}_tests.log; exit 1; }
  tail -2 gpurun_out/${tag}_tests.log
fi
for wl in "$@"; do
  timeout -k 10 400 python3 bench.py --workload $wl --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-other-workloads > gpurun_out/${tag}_${wl}.json 2> gpurun_out/${tag}_${wl}.err || { tail -30 gpurun_out/${tag}_${wl}.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_${wl}.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'])"
done
