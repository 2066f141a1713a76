#!/bin/bash
# what the eager probe iteration (the roofline's event-timed sample) costs the timed region
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 400 python3 bench.py --steps 40 --warmup 5 --no-other-workloads --no-cpu-baseline "$@" > gpurun_out/pc_$tag.json 2> gpurun_out/pc_$tag.err; python3 -c "
import json
for l in open('gpurun_out/pc_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
run every10 --probe-every 10
run every40 --probe-every 40
run every5 --probe-every 5
