#!/bin/bash
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
tag=${1:-r3b}
timeout -k 10 700 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -30 gpurun_out/${tag}_bench.err; exit 1; }
python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads([l for l in open("gpurun_out/%s_bench.json" % tag).read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
for k, v in d.get("other_workloads", {}).items():
    print("  ", k, v.get("value"), v.get("ms_per_step"), v.get("error"), (v.get("cpu_baseline") or {}).get("value"))
print("  cpu", d.get("cpu_baseline"), d.get("cpu_baseline_8_threads"))
PY
