#!/bin/bash
# bench.py under a few MIOpen solver-selection settings (GPU box).  Usage: scripts/try_env.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
run() { tag=$1; shift; echo "== $tag"; env "$@" python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
run default A=1
run no_igemm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0
run no_igemm_asm MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0
echo "== channels_last"; python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --channels-last 2>&1 | tail -1 | cut -c1-200
