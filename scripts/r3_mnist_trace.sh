#!/bin/bash
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
w=${1:-mnist_ee_at}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads > gpurun_out/r3f_${w}_profiled.json 2> gpurun_out/r3f_${w}_prof.err
t=$(find /tmp/prof_$w -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_sequence.py "$t" > gpurun_out/r3f_${w}_trace_sequence.txt
python3 scripts/trace_breakdown.py "$t" 100 40 > gpurun_out/r3f_${w}_trace_breakdown.txt
head -40 gpurun_out/r3f_${w}_trace_sequence.txt
