#!/bin/bash
# round 3, first GPU call: the GPU suite, the default bench line (with the other workloads), the free-AT workload on its own + its kernel table
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3a_gputests.log 2>&1 || { tail -40 gpurun_out/r3a_gputests.log; exit 1; }
tail -3 gpurun_out/r3a_gputests.log
timeout -k 10 600 python3 bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err || { tail -30 gpurun_out/r3a_bench.err; exit 1; }
echo "bench done"
timeout -k 10 400 python3 bench.py --workload imagenet_free_at --steps 10 --warmup 2 > gpurun_out/r3a_freeat.json 2> gpurun_out/r3a_freeat.err || { tail -30 gpurun_out/r3a_freeat.err; exit 1; }
echo "freeat done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fa -- python3 bench.py --workload imagenet_free_at --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3a_freeat_profiled.json 2> gpurun_out/r3a_freeat_prof.err
f=$(find /tmp/prof_fa -name "*kernel_stats.csv" | head -1)
python3 scripts/summarize_stats.py "$f" 70 > gpurun_out/r3a_freeat_kernel_stats.txt
t=$(find /tmp/prof_fa -name "*kernel_trace.csv" | head -1)
python3 scripts/trace_breakdown.py "$t" 200 60 > gpurun_out/r3a_freeat_trace_breakdown.txt
python3 - <<'PY'
import json
for f in ("gpurun_out/r3a_bench.json", "gpurun_out/r3a_freeat.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], {k: (v.get("value"), v.get("error")) for k, v in d.get("other_workloads", {}).items()}, d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_8_threads", {}).get("value"))
PY
