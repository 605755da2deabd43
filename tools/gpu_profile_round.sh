#!/bin/bash
# The round's tracked profiles (run on the GPU box through gpurun; copies land in gpurun_out/, to be moved into profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (short run)          -> <tag>_kernel_stats.csv + the JSON line of that run
#   2. SQ counter passes (MFMA utilisation) of the f16c generator leg                      -> <tag>_pmc_mfma.json      (tools/gpu_pmc_mfma.sh)
#   3. FETCH_SIZE / WRITE_SIZE passes: generator leg and the GeM-ResNet-101 leg            -> <tag>_pmc_traffic*.json  (profiles/summarise_pmc.py)
TAG=${1:-r03}
R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
echo "== kernel stats"; (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
echo "== SQ passes"; bash tools/gpu_pmc_mfma.sh $TAG > $O/pmc_sq.log 2>&1; cp $R/gpurun_out/pmc_mfma_$TAG/pmc_mfma.json $O/${TAG}_pmc_mfma.json
echo "== traffic passes (generator)"
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fg --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact > $O/fg.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wg --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact > $O/wg.log 2>&1)
python3 profiles/summarise_pmc.py $(ls $O/fg/*/*counter_collection.csv | head -1) $(ls $O/wg/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact (f16c generator, 64x3x256x256)"
echo "== traffic passes (GeM-ResNet-101)"
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/fr.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/wr.log 2>&1)
python3 profiles/summarise_pmc.py $(ls $O/fr/*/*counter_collection.csv | head -1) $(ls $O/wr/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic_r101.json "tools/r101_forward.py 3: GeM-ResNet-101 forward, 32x3x1024x1024, fp16 mode (sum over the kernels of one forward = launches x bytes / number of forwards)"
ls -la $O
