#!/bin/bash
# round-4 tracked profiles, part B (after the fused 3x3 + expand launch of ResNet-101 layer3): kernel stats of the default bench + its line, the un-profiled bench line,
# FETCH / WRITE passes of the GeM-ResNet-101 forward
TAG=r04
R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_r04b
mkdir -p $O
python3 bench.py --steps 100 > $O/${TAG}_bench_line.json 2> $O/bench.err || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err) || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/fr.log 2>&1) || exit 1
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/wr.log 2>&1) || exit 1
python3 profiles/summarise_pmc.py $(ls $O/fr/*/*counter_collection.csv | head -1) $(ls $O/wr/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic_r101.json "tools/r101_forward.py 3: GeM-ResNet-101 forward, 32x3x1024x1024, fp16 mode, round-4 build (layer3: 3x3 + expand + residual in one launch)"
rm -rf $O/stats $O/fr $O/wr
ls -la $O
