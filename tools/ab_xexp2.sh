#!/bin/bash
# dev: fused 3x3 + expand launch: 16-row patches (one workgroup per CU) vs 8-row patches (two per CU) at several start-up staggers
O=gpurun_out
for cfg in "16 0" "8 0" "8 15" "8 30" "8 -30" "8 30" "16 0"; do
  set -- $cfg
  echo "== PH $1 stagger $2" >> $O/xexp2.log
  GDT_XEXP_PH=$1 GDT_XEXP_STAGGER_US=$2 python tools/r101_ops.py 2>&1 | grep -E "variant  937|total" | tail -3 >> $O/xexp2.log
  GDT_XEXP_PH=$1 GDT_XEXP_STAGGER_US=$2 python bench.py --steps 30 --no-cpu-baseline --no-fast --no-exact 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('bench secondary', d['secondary']['value'], d['secondary']['ms_per_step'])" >> $O/xexp2.log
done
cat $O/xexp2.log
