#!/bin/bash
# headline workload with MIOpen's weight-gradient solver families switched off one at a time (what does the update phase pay for igemm + transposes?)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 --no-other-workloads --no-cpu-baseline > gpurun_out/wrwenv_$tag.json 2> gpurun_out/wrwenv_$tag.err; python3 -c "
import json,sys
for l in open('gpurun_out/wrwenv_$tag.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$tag', d['value'], d['ms_per_step'])"; }
run default A=1
run noigemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
run noigemm_wrw MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
