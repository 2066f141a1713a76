set -e
cd "${GRAFT_REPO_ROOT:-.}"
bash scripts/prof_round.sh round2_g > gpurun_out/prof_round.log 2>&1
echo "# python bench.py --no-cpu-baseline --workload <w>, one MI355X, round 2's final build (the default workload is tiny_ee_at; round 1: 5554 / - / ~5000 / ~5700 img/s)" > gpurun_out/round2_g_workloads.txt
for w in tiny_ee_at tiny_at tiny_trades mnist_ee_at; do
  python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-12s %9.1f img/s  %8.3f ms/step   %s' % ('$w', d['value'], d['ms_per_step'], d['config']['workload']))" >> gpurun_out/round2_g_workloads.txt
done
cat gpurun_out/round2_g_workloads.txt
