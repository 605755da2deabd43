#!/bin/bash
# round-4 tracked profiles, part A (run through gpurun; copies land in gpurun_out/prof_r04, to be moved into profiles/):
#   kernel stats of the default bench under rocprofv3 + its JSON line; SQ counter passes (MFMA utilisation) of the f16c generator leg;
#   FETCH_SIZE / WRITE_SIZE passes of the generator leg
TAG=r04
R=$(pwd)
export TMPDIR=/tmp
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
echo "== kernel stats"; (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/stats.err)
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
echo "== SQ passes"; bash tools/gpu_pmc_mfma.sh $TAG > $O/pmc_sq.log 2>&1; cp $R/gpurun_out/pmc_mfma_$TAG/pmc_mfma.json $O/${TAG}_pmc_mfma.json
echo "== traffic passes (generator)"
(cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fg --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact > $O/fg.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wg --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact > $O/wg.log 2>&1)
python3 profiles/summarise_pmc.py $(ls $O/fg/*/*counter_collection.csv | head -1) $(ls $O/wg/*/*counter_collection.csv | head -1) $O/${TAG}_pmc_traffic.json "bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact (f16c generator, 64x3x256x256), round-4 build (conv3x3_halo_c16)"
rm -rf $O/stats $O/fg $O/wg $R/gpurun_out/pmc_mfma_$TAG/p*
ls -la $O
