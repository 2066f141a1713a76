#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_chain.py -x -q -m gpu > gpurun_out/r4i_chain_tests.log 2>&1 || { tail -40 gpurun_out/r4i_chain_tests.log; exit 1; }
tail -2 gpurun_out/r4i_chain_tests.log
echo "--- EEADV_CHAIN_BANDS=1" > gpurun_out/r4i_chain_bench.txt
EEADV_CHAIN_BANDS=1 python3 scripts/chain_bench.py 100x3x64x64,1600x3x64x64,50x1x28x28 2>&1 | grep "chain_fwd" >> gpurun_out/r4i_chain_bench.txt
echo "--- EEADV_CHAIN_BANDS=0" >> gpurun_out/r4i_chain_bench.txt
EEADV_CHAIN_BANDS=0 python3 scripts/chain_bench.py 100x3x64x64,1600x3x64x64,50x1x28x28 2>&1 | grep "chain_fwd" >> gpurun_out/r4i_chain_bench.txt
cat gpurun_out/r4i_chain_bench.txt
