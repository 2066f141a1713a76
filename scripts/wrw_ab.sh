#!/bin/bash
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 300 python3 scripts/wrw_probe.py 100 > gpurun_out/wrw_probe.txt 2>&1 || { tail -20 gpurun_out/wrw_probe.txt; exit 1; }
cat gpurun_out/wrw_probe.txt
run() { tag=$1; shift; env "$@" timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-other-workloads --no-cpu-baseline > gpurun_out/wrwb_$tag.json 2> gpurun_out/wrwb_$tag.err; python3 -c "
import json
for l in open('gpurun_out/wrwb_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])"; }
run own A=1
run stock EEADV_STOCK_WRW=1
