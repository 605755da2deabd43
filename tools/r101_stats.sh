#!/bin/bash
# dev (GPU box): per-kernel averages of six GeM-ResNet-101 forwards (32 x 1024^2) for one library build and environment
#   tools/r101_stats.sh <tag> [lib] [ENV=val ...]
TAG=$1; LIB=$2; shift 2
export TMPDIR=/tmp
R=$PWD
for kv in "$@"; do export "$kv"; done
[ -n "$LIB" ] && [ "$LIB" != "-" ] && export GANDTR_HIP_LIB=$R/$LIB
mkdir -p $R/gpurun_out/ks_$TAG
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$TAG --output-format csv -- python3 $R/tools/r101_forward.py 6 > $R/gpurun_out/ks_$TAG/log.txt 2>&1)
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/ks_$TAG/*/*kernel_stats.csv")[0]
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    ms = float(r["TotalDurationNs"]) / 6e6; tot += ms
    if ms > 0.05: print("%-62s %4d x %7.1f us = %6.3f ms" % (n, int(r["Calls"]) // 6, float(r["AverageNs"]) / 1e3, ms))
print("$TAG total kernel time per forward %.3f ms" % tot)
PY
rm -rf $R/gpurun_out/ks_$TAG
