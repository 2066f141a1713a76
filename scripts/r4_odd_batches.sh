set -e
cd "${GRAFT_REPO_ROOT:-.}"
for b in 1 7 33 256; do
  timeout -k 10 300 python3 bench.py --workload tiny_ee_at --batch $b --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny_ee_at batch', $b, d['value'], d['final_loss'])"
done
for b in 1 13; do
  timeout -k 10 300 python3 bench.py --workload mnist_ee_at --batch $b --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mnist_ee_at batch', $b, d['value'], d['final_loss'])"
done
timeout -k 10 300 python3 bench.py --workload tiny_trades --batch 24 --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiny_trades batch 24', d['value'], d['final_loss'])"
