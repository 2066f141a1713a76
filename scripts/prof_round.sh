#!/bin/bash
# One round's measurement set on the GPU box (everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/):
#   1. the default bench line (with the other BASELINE configs and the cpu baselines) -> <tag>_bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command (no cpu baseline)  -> <tag>_kernel_stats.txt, <tag>_rocprof_kernel_us.json
#   3. steady-state breakdown of that trace (last 200 ms)                      -> <tag>_trace_breakdown.txt
#   4. HBM traffic of the front-end kernels and of the 3x3 convolution kernels at the bench shape, two --pmc passes each
#      (FETCH_SIZE, WRITE_SIZE; counters only)
#                                                                              -> <tag>_pmc_traffic.txt, <tag>_pmc_traffic.json
# Usage: scripts/prof_round.sh <tag>
set -e
tag=$1
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 bench.py --no-cpu-baseline --no-other-workloads > gpurun_out/${tag}_bench_profiled.json 2> gpurun_out/${tag}_prof.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_stats.py "$f" 60 > gpurun_out/${tag}_kernel_stats.txt
python3 scripts/export_rocprof.py "$f" gpurun_out/${tag}_rocprof_kernel_us.json "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-workloads ($tag)" > /dev/null
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$t" 200 60 > gpurun_out/${tag}_trace_breakdown.txt
echo "kernel trace done"
# BASELINE config 5 on its own: the line and its kernel table
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa_$tag -- python3 bench.py --workload imagenet_free_at --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_free_at_profiled.json 2> gpurun_out/${tag}_free_at_prof.err
f=$(find /tmp/prof_fa_$tag -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_stats.py "$f" 60 > gpurun_out/${tag}_free_at_kernel_stats.txt
t=$(find /tmp/prof_fa_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$t" 200 60 > gpurun_out/${tag}_free_at_trace_breakdown.txt
echo "free-AT trace done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  CHAIN_BENCH_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -- python3 scripts/chain_bench.py 100x3x64x64 > gpurun_out/${tag}_pmc_$ctr.log 2>&1
  PROBE_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_wino_$ctr -- python3 scripts/wino_probe.py 100 > gpurun_out/${tag}_pmc_wino_$ctr.log 2>&1
  PROBE_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_s2_$ctr -- python3 scripts/s2_probe.py 100 > gpurun_out/${tag}_pmc_s2_$ctr.log 2>&1
done
rm -f gpurun_out/${tag}_pmc_traffic.json
python3 scripts/pmc_to_json.py /tmp/pmc_${tag}_FETCH_SIZE /tmp/pmc_${tag}_WRITE_SIZE 100x3x64x64 gpurun_out/${tag}_pmc_traffic.json \
  "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over scripts/chain_bench.py 100x3x64x64 (plain launches); KiB counters, FETCH doubled per guides/MI355X_MICROARCH.md" > gpurun_out/${tag}_pmc_traffic.txt
python3 scripts/pmc_to_json.py /tmp/pmc_${tag}_wino_FETCH_SIZE /tmp/pmc_${tag}_wino_WRITE_SIZE 100x3x64x64 gpurun_out/${tag}_pmc_traffic.json >> gpurun_out/${tag}_pmc_traffic.txt
python3 scripts/pmc_to_json.py /tmp/pmc_${tag}_s2_FETCH_SIZE /tmp/pmc_${tag}_s2_WRITE_SIZE 100x3x64x64 gpurun_out/${tag}_pmc_traffic.json >> gpurun_out/${tag}_pmc_traffic.txt
t=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_sequence.py "$t" > gpurun_out/${tag}_trace_sequence.txt
echo "pmc done"
cut -c1-600 gpurun_out/${tag}_bench.json
cat gpurun_out/${tag}_pmc_traffic.txt
