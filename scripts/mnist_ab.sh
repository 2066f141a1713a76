#!/bin/bash
# BASELINE config 2 (MNIST EE AT) with and without the hand-written parameter gradients
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 300 python3 bench.py --workload mnist_ee_at --steps 60 --warmup 5 --no-cpu-baseline --no-other-workloads > gpurun_out/mnab_$tag.json 2> gpurun_out/mnab_$tag.err; python3 -c "
import json
for l in open('gpurun_out/mnab_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])"; }
run own A=1
run stock EEADV_STOCK_WRW=1
run own2 A=1
