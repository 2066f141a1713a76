#!/bin/bash
# SQ counters of the MFMA convolution kernel (separate --pmc passes).  Usage: scripts/prof_pmc_conv.sh
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $ctrs | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmcconv_$tag -- python3 scripts/conv_probe.py > gpurun_out/pmcconv_$tag.log 2>&1 || true
  f=$(find /tmp/pmcconv_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "conv3x3s1" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-28s %14.0f per launch (%d launches)" % (k, v / n, n))
PY
done
