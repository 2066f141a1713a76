#!/bin/bash
# round 4, first GPU call: the fused eval-mode kernels against the unfused sequence, then TRADES (its attack runs in eval mode) with and without them
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_evalfuse.py -x -q > gpurun_out/r4a_evalfuse_tests.log 2>&1 || { tail -60 gpurun_out/r4a_evalfuse_tests.log; exit 1; }
tail -3 gpurun_out/r4a_evalfuse_tests.log
timeout -k 10 300 python3 bench.py --workload tiny_trades --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4a_trades_fused.json 2> gpurun_out/r4a_trades_fused.err || { tail -30 gpurun_out/r4a_trades_fused.err; exit 1; }
EEADV_STOCK_GLUE=evalfuse timeout -k 10 300 python3 bench.py --workload tiny_trades --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4a_trades_unfused.json 2> gpurun_out/r4a_trades_unfused.err || { tail -30 gpurun_out/r4a_trades_unfused.err; exit 1; }
python3 - <<'PY'
import json
for t in ("fused", "unfused"):
    d = json.loads(open("gpurun_out/r4a_trades_%s.json" % t).read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"])
PY
