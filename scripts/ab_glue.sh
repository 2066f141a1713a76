#!/bin/bash
# A/B of the fused CNN-glue kernels on one box: bench.py with each piece routed back to the stock ops (EEADV_STOCK_GLUE).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
run() { echo -n "stock=[$1] "; EEADV_STOCK_GLUE=$1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for v in "$@"; do run "$v"; done
