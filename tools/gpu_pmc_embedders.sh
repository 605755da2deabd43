#!/bin/bash
# MFMA-utilisation counters of the descriptor legs (GeM-ResNet-101 / GeM-VGG16 forwards, 32 x 1024^2): two SQ passes each
#   gpurun -- bash tools/gpu_pmc_embedders.sh
R=$(pwd)
export TMPDIR=/tmp
for NET in ${NETS:-r101 vgg16}; do
  OUT=$R/gpurun_out/pmc_$NET
  mkdir -p $OUT
  SCRIPT=$R/tools/r101_forward.py
  [ $NET == vgg16 ] && SCRIPT=$R/tools/vgg16_forward.py
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i --output-format csv -- python3 $SCRIPT 3 > $OUT/p$i.log 2>&1) || echo "pass $i failed" >> $OUT/fail.log
  done
  python3 $R/profiles/summarise_pmc_sq.py $OUT $R/gpurun_out/${TAG:-r04}_pmc_mfma_$NET.json conv3x3_expand_rb_kernel conv3x3_halo_rb_kernel conv1x1_rb_kernel conv_bneck_kernel conv_stem_pair conv3x3_halo_kernel || true
  rm -rf $OUT/p1 $OUT/p2
done
