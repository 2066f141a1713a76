#!/bin/bash
# like ab_env.sh for another workload.  Usage: scripts/ab_env_w.sh <workload> "A=1" "B=2" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
w=$1; shift
for v in "$@"; do
  echo -n "[$v] "
  env $v python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
