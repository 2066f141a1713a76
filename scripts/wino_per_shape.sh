#!/bin/bash
# round 4: the three Winograd kernels at the bench shapes, per shape: time (un-profiled probe + rocprofv3 kernel trace), frac /
# frac_executed against the f32 MFMA peak, HBM traffic (--pmc FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ counters
# (one more counters-only pass).  Output: gpurun_out/<tag>_wino_per_shape.txt
set -e
tag=${1:-r4w}
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
mkdir -p gpurun_out
python3 scripts/wino_probe.py 100 > gpurun_out/${tag}_wino_probe.txt 2>/dev/null
cd /tmp
PROBE_EAGER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws_trace_$tag -- python3 $R/scripts/wino_probe.py 100 > /dev/null 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  PROBE_EAGER=1 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/ws_${ctr}_$tag -- python3 $R/scripts/wino_probe.py 100 > /dev/null 2>&1
done
PROBE_EAGER=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/ws_sq_$tag -- python3 $R/scripts/wino_probe.py 100 > /dev/null 2>&1
cd $R
python3 - $tag > gpurun_out/${tag}_wino_per_shape.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
def counters(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "wino3x3_" in r["Kernel_Name"]:
            k = (r["Kernel_Name"], r["Counter_Name"])
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}
stats = {}
f = glob.glob("/tmp/ws_trace_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "wino3x3_" in r["Name"]:
        stats[r["Name"]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
fe, wr, sq = counters("/tmp/ws_FETCH_SIZE_" + tag), counters("/tmp/ws_WRITE_SIZE_" + tag), counters("/tmp/ws_sq_" + tag)
GF, PEAK = 1.8874368, 157.3
alg = {"16": (2 * 100 * 64 * 256 * 4 + 16 * 64 * 64 * 4) / 1e6, "8,": (2 * 100 * 128 * 64 * 4 + 16 * 128 * 128 * 4) / 1e6, "map4": (2 * 100 * 256 * 16 * 4 + 16 * 256 * 256 * 4) / 1e6}
print("# scripts/wino_per_shape.sh: the plain Winograd kernels at batch 100 (ResNet-18 layers 1 / 2 / 3), one MI355X; 1.887 GFLOP of convolution per launch,")
print("# f32 MFMA peak 157.3 TFLOP/s, the matrix cores execute 4/9 of the flops; traffic = FETCH_SIZE * 2 + WRITE_SIZE (KiB counters, gfx950 correction);")
print("# algorithmic bytes = input + output + transformed filters once.  SQ counters are sums over the launch (all CUs).")
print(open("gpurun_out/%s_wino_probe.txt" % tag).read().rstrip())
print()
for name, (us, calls) in sorted(stats.items()):
    key = "map4" if "map4" in name else ("16" if "pc_kernel<16" in name else "8,")
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    tr = (fe.get((name, "FETCH_SIZE"), 0) * 2 + wr.get((name, "WRITE_SIZE"), 0)) * 1024 / 1e6
    w, busy = sq.get((name, "SQ_WAVE_CYCLES"), 0), sq.get((name, "SQ_VALU_MFMA_BUSY_CYCLES"), 0)
    print("%s\n    rocprofv3 %.2f us (%d launches)  frac %.3f  frac_executed %.3f  traffic %.2f MB = %.2f x algorithmic (%.2f MB)" % (
        short, us, calls, GF * 1e9 / (us * 1e-6) / 1e12 / PEAK, GF * 4 / 9 * 1e9 / (us * 1e-6) / 1e12 / PEAK, tr, tr / alg[key], alg[key]))
    print("    SQ_VALU_MFMA_BUSY_CYCLES %.3g = %.1f %% of (cycles x 4 SIMDs x CUs used: duration x 2.4 GHz x 4 x %d)   wave cycles %.3g, waiting %.0f %%, LDS bank conflict %.1f %% of LDS active" % (
        busy, 100 * busy / (us * 1e-6 * 2.4e9 * 4 * 200), 200, w, 100 * sq.get((name, "SQ_WAIT_ANY"), 0) / max(w, 1),
        100 * sq.get((name, "SQ_LDS_BANK_CONFLICT"), 0) / max(sq.get((name, "SQ_LDS_IDX_ACTIVE"), 1), 1)))
PY
cat gpurun_out/${tag}_wino_per_shape.txt
