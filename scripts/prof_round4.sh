#!/bin/bash
# Round 4's measurement set: scripts/prof_round.sh <tag> (default line, rocprofv3 stats / trace of it, free-AT, TCC traffic) plus
#   - kernel traces of the eval-mode workload (validate(): PGD-50) and of TRADES -> <tag>_eval_*.txt, <tag>_trades_*.txt
#   - the EE front end at 224 x 224 under rocprofv3 --kernel-trace --stats    -> <tag>_frontend224_*.txt
# Usage: scripts/prof_round4.sh <tag>
set -e
tag=$1
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
bash scripts/prof_round.sh $tag
for wl in tiny_ee_eval_pgd50 tiny_trades mnist_ee_at; do
  short=${wl#tiny_}; short=${short%%_*}
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_$wl -- python3 $R/bench.py --workload $wl --steps 8 --warmup 2 --no-cpu-baseline --no-other-workloads > $R/gpurun_out/${tag}_${wl}_profiled.json 2> $R/gpurun_out/${tag}_${wl}_prof.err
  cd $R
  t=$(find /tmp/prof_${tag}_$wl -name "*kernel_trace.csv" | head -1)
  python3 scripts/trace_breakdown.py "$t" 200 60 > gpurun_out/${tag}_${wl}_trace_breakdown.txt
  python3 scripts/trace_sequence.py "$t" $([ $wl = tiny_trades ] && echo stem_fwd_mfma_kernel || echo chain_fwd_kernel) > gpurun_out/${tag}_${wl}_trace_sequence.txt || true
  echo "$wl trace done"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag}_fe224 -- python3 $R/scripts/frontend224_probe.py > $R/gpurun_out/${tag}_frontend224.txt 2> $R/gpurun_out/${tag}_frontend224.err
cd $R
f=$(find /tmp/prof_${tag}_fe224 -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_stats.py "$f" 12 >> gpurun_out/${tag}_frontend224.txt
cat gpurun_out/${tag}_frontend224.txt
