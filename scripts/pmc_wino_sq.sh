#!/bin/bash
# SQ counters of the Winograd kernels at the bench shapes (one pass: 8 SQ slots), plain launches
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
PROBE_EAGER=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_wino_sq -- python3 scripts/wino_probe.py 100 > gpurun_out/r3i_pmc_wino_sq.log 2>&1
python3 scripts/pmc_kernels.py /tmp/pmc_wino_sq wino3x3 > gpurun_out/r3i_pmc_wino_sq.txt
cat gpurun_out/r3i_pmc_wino_sq.txt
