#!/bin/bash
# dev: start-up stagger of conv3x3_halo_c (GDT_C_STAGGER_US) on the generator's shift forms (stride-2 / transposed: 4-16 tiles per workgroup)
O=gpurun_out
for s in 0 3 6 12 0 6; do
  echo "== GDT_C_STAGGER_US $s" >> $O/stag_shift.log
  GDT_C_STAGGER_US=$s python tools/gen_ops.py 2>&1 | grep -E "variant  (98|99)0|total" >> $O/stag_shift.log
done
cat $O/stag_shift.log
