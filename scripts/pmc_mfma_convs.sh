#!/bin/bash
# HBM traffic (TCC FETCH_SIZE / WRITE_SIZE, separate passes, counters only) of the Winograd and stride-2 kernels at the bench shapes.
# Usage: scripts/pmc_mfma_convs.sh <tag>  ->  gpurun_out/<tag>_pmc_convs.txt
set -e
tag=$1
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  PROBE_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcc_${tag}_wino_$ctr -- python3 scripts/wino_probe.py 100 > gpurun_out/${tag}_pmcc_wino_$ctr.log 2>&1
  PROBE_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcc_${tag}_s2_$ctr -- python3 scripts/s2_probe.py 100 > gpurun_out/${tag}_pmcc_s2_$ctr.log 2>&1
done
rm -f gpurun_out/${tag}_pmc_convs.json
python3 scripts/pmc_to_json.py /tmp/pmcc_${tag}_wino_FETCH_SIZE /tmp/pmcc_${tag}_wino_WRITE_SIZE 100x3x64x64 gpurun_out/${tag}_pmc_convs.json > gpurun_out/${tag}_pmc_convs.txt
python3 scripts/pmc_to_json.py /tmp/pmcc_${tag}_s2_FETCH_SIZE /tmp/pmcc_${tag}_s2_WRITE_SIZE 100x3x64x64 gpurun_out/${tag}_pmc_convs.json >> gpurun_out/${tag}_pmc_convs.txt
cat gpurun_out/${tag}_pmc_convs.txt
