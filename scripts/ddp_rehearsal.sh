#!/bin/bash
# round 3: the multi-rank update on a one-GPU box - two gloo ranks time-sharing the GPU, then ONE rank over RCCL with the collectives forced on
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_ddp.py tests/test_syncbn.py tests/test_awp.py tests/test_gpu_attacks.py -m gpu -x -q -k "segmented or syncbn or awp or free_at" > gpurun_out/r3d_tests.log 2>&1 || { tail -40 gpurun_out/r3d_tests.log; exit 1; }
tail -3 gpurun_out/r3d_tests.log
timeout -k 10 500 python3 -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 scripts/ddp_same_gpu.py > gpurun_out/r3d_ddp_gloo2.log 2>&1 || { tail -40 gpurun_out/r3d_ddp_gloo2.log; exit 1; }
grep -E "^rank|phases" gpurun_out/r3d_ddp_gloo2.log
DDP_BACKEND=nccl EEADV_FORCE_COLLECTIVES=1 timeout -k 10 400 python3 -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 scripts/ddp_same_gpu.py > gpurun_out/r3d_ddp_rccl1.log 2>&1 || { tail -40 gpurun_out/r3d_ddp_rccl1.log; exit 1; }
grep -E "^rank|phases" gpurun_out/r3d_ddp_rccl1.log
