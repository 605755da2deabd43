#!/bin/bash
# dev (GPU box): per-kernel time of the device JPEG decoder on the tools/jpeg_bench.py workload
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/jpeg_prof
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/jpeg_prof --output-format csv -- python3 $R/tools/jpeg_bench.py > $R/gpurun_out/jpeg_prof/bench.json 2> $R/gpurun_out/jpeg_prof/err.txt)
cp $(ls $R/gpurun_out/jpeg_prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/jpeg_kernel_stats.csv
rm -rf $R/gpurun_out/jpeg_prof/*/
head -14 $R/gpurun_out/jpeg_kernel_stats.csv | cut -c1-200
