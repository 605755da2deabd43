#!/bin/bash
# round-3 final profiles: kernel stats of the default bench, FETCH/WRITE passes of the GeM-ResNet-101 forward
TAG=r03
R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_final
mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/fr.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/wr.log 2>&1)
python3 profiles/summarise_pmc.py $(ls $O/fr/*/*counter_collection.csv | head -1) $(ls $O/wr/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic_r101.json "tools/r101_forward.py 3: GeM-ResNet-101 forward, 32x3x1024x1024, fp16 mode, final round-3 build (stem + max-pool from the fp32 image, projection shortcuts folded into the expand convs, layer1/2 Bottlenecks fused)"
rm -rf $O/stats $O/fr $O/wr
ls -la $O
