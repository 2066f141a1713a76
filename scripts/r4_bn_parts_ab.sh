set -e
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_chain.py tests/test_gpu_trainfuse.py -x -q -m gpu -k "bn or batch or reproducible or trainfuse or classifier" 2>&1 | tail -2
for i in 1 2; do
for v in 0 1; do
  EEADV_BN_PARTS=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EEADV_BN_PARTS=$v', d['value'], d['ms_per_step'])"
done
done
