#!/bin/bash
# round 4: run-to-run spread of the bench legs on ONE box: five back-to-back runs of each workload (20 steps, 5 warm-up)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for wl in tiny_ee_at tiny_trades mnist_ee_at tiny_ee_eval_pgd50; do
  vals=""
  for i in 1 2 3 4 5; do
    v=$(timeout -k 10 300 python3 bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    vals="$vals $v"
  done
  python3 -c "
import sys
v=[float(x) for x in sys.argv[2:]]
m=sum(v)/len(v)
print('%-20s runs %s   mean %.1f  min %.1f  max %.1f  spread %.2f %%' % (sys.argv[1], ' '.join('%.0f' % x for x in v), m, min(v), max(v), 100*(max(v)-min(v))/m))" $wl $vals
done
