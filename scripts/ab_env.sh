#!/bin/bash
# bench.py under a list of VAR=value settings, one run each, on one box.  Usage: scripts/ab_env.sh "A=1" "B=2 C=3" ...
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for v in "$@"; do
  echo -n "[$v] "
  env $v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
