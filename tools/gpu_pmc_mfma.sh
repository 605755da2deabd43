#!/bin/bash
# rocprofv3 counter passes over the f16c generator leg of bench.py; run on the GPU box through gpurun:
#   gpurun -- bash tools/gpu_pmc_mfma.sh <tag> [groups file]
# One pass per counter group (the guide: counters in their own run with --kernel-trace only), the program directly after --.
# Default groups = the MFMA-utilisation evidence of profiles/rNN_pmc_mfma.json; a groups file holds one group per line.
TAG=${1:-r03}
GROUPS_FILE=$2
R=$(pwd)
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_mfma_$TAG
mkdir -p $OUT
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-fast --no-exact"
if [ -z "$GROUPS_FILE" ]; then
  GROUPS_FILE=$OUT/groups.txt
  cat > $GROUPS_FILE <<'EOG'
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES
SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA
EOG
fi
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i --output-format csv -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1) || echo "pass $i ($grp) failed" >> $OUT/fail.log
  echo "pass $i done: $grp"
done < $GROUPS_FILE
python3 $R/profiles/summarise_pmc_sq.py $OUT $OUT/pmc_mfma.json || true
