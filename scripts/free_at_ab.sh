#!/bin/bash
# BASELINE config 5 with and without ee_wrw.hip (only the stem's weight gradient is ours at 224x224)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 500 python3 bench.py --workload imagenet_free_at --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/faab_$tag.json 2> gpurun_out/faab_$tag.err; python3 -c "
import json
for l in open('gpurun_out/faab_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])"; }
run own A=1
run stock EEADV_STOCK_WRW=1
run own2 A=1
