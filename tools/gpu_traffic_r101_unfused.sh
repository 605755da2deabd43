R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/prof_r03u; mkdir -p $O
(cd /tmp && GDT_CONV_BNECK=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/fr.log 2>&1)
(cd /tmp && GDT_CONV_BNECK=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/wr --output-format csv -- python3 $R/tools/r101_forward.py 3 > $O/wr.log 2>&1)
python3 profiles/summarise_pmc.py $(ls $O/fr/*/*counter_collection.csv | head -1) $(ls $O/wr/*/*counter_collection.csv | head -1) $O/r03_pmc_traffic_r101_unfused.json "GDT_CONV_BNECK=0 tools/r101_forward.py 3 (layer-by-layer Bottlenecks)"
