#!/bin/bash
# dev (GPU box): back-to-back A/B of library builds on ONE box (boxes of the pool differ by +-3 %):  tools/ab.sh <rounds> <lib> [<lib> ...] [-- bench args]
# prints images/s, ms/step and the dominant kernel's average launch time per run
R=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ARGS=${@:---no-cpu-baseline --no-secondary --no-fast --no-exact}
for r in $(seq 1 $R); do
  for L in "${LIBS[@]}"; do
    GANDTR_HIP_LIB=$PWD/$L python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
k=d['roofline']['all_conv_kernels']
print('$L', d['value'], d['ms_per_step'], 'dominant', d['roofline']['avg_launch_ms'], {a.replace('conv3x3_halo_c_kernel','c').replace('conv_',''):b['ms_per_step'] for a,b in k.items()})
"
  done
done
