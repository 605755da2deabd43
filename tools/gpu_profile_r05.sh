#!/bin/bash
# round-5 tracked profiles (run through gpurun; results land in gpurun_out/prof_r05, to be copied into profiles/):
#   kernel stats of the default bench under rocprofv3 + its JSON line; FETCH_SIZE / WRITE_SIZE passes of the f16c generator leg and of the GeM-ResNet-101
#   forward (layer3 as a chain: the next block's reduce conv inside the fused launch); SQ counter passes (MFMA utilisation) of both
TAG=r05
R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
echo "== kernel stats"
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err) || exit 1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
echo "== traffic passes (generator)"
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fg --output-format csv -- python3 $R/bench.py $ARGS > $O/fg.log 2>&1) || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wg --output-format csv -- python3 $R/bench.py $ARGS > $O/wg.log 2>&1) || exit 1
python3 profiles/summarise_pmc.py $(ls $O/fg/*/*counter_collection.csv | head -1) $(ls $O/wg/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic.json "bench.py $ARGS (f16c generator, 64x3x256x256), round-5 build"
echo "== traffic passes (GeM-ResNet-101)"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/fr.log 2>&1) || exit 1
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/wr.log 2>&1) || exit 1
python3 profiles/summarise_pmc.py $(ls $O/fr/*/*counter_collection.csv | head -1) $(ls $O/wr/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic_r101.json "tools/r101_forward.py 3: GeM-ResNet-101 forward, 32x3x1024x1024, fp16 mode, round-5 build (layer3: 3x3 + expand + residual + the next block's reduce conv in one launch)"
echo "== SQ passes"
bash tools/gpu_pmc_mfma.sh $TAG > $O/pmc_sq.log 2>&1; cp $R/gpurun_out/pmc_mfma_$TAG/pmc_mfma.json $O/${TAG}_pmc_mfma.json
TAG=$TAG NETS=r101 bash tools/gpu_pmc_embedders.sh > $O/pmc_sq_r101.log 2>&1; cp $R/gpurun_out/${TAG}_pmc_mfma_r101.json $O/ 2>/dev/null
rm -rf $O/stats $O/fg $O/wg $O/fr $O/wr $R/gpurun_out/pmc_mfma_$TAG/p*
ls -la $O
