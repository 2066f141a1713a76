#!/bin/bash
# round 4: the whole -m gpu suite + smoke, then a kernel trace of the eval-mode workload (validate(): PGD-50 in eval mode) -> sequence / breakdown
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
R=$PWD
tag=${1:-r4b}
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gputests.log 2>&1 || { tail -40 gpurun_out/${tag}_gputests.log; exit 1; }
tail -3 gpurun_out/${tag}_gputests.log
python3 __graft_entry__.py smoke 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --workload tiny_ee_eval_pgd50 --steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads > $R/gpurun_out/${tag}_eval_bench_profiled.json 2> $R/gpurun_out/${tag}_eval_prof.err || { tail -30 $R/gpurun_out/${tag}_eval_prof.err; exit 1; }
cd $R
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$t" 200 70 > gpurun_out/${tag}_eval_trace_breakdown.txt
python3 scripts/trace_sequence.py "$t" > gpurun_out/${tag}_eval_trace_sequence.txt
head -3 gpurun_out/${tag}_eval_trace_breakdown.txt
head -2 gpurun_out/${tag}_eval_trace_sequence.txt
timeout -k 10 300 python3 bench.py --workload tiny_ee_eval_pgd50 --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/${tag}_eval_bench.json 2> gpurun_out/${tag}_eval_bench.err
python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_eval_bench.json').read().strip().splitlines()[-1]); print('eval pgd50', d['value'], d['ms_per_step'])"
