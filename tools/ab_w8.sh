#!/bin/bash
# dev: A/B of the 1 x 8 (two waves per SIMD) layout of conv3x3_halo_c against the shipped 1 x 4 one and of build variants (tmpbin/lib_*.so), inside one gpurun call
mkdir -p gpurun_out
B="python bench.py --steps 50 --warmup 10 --no-fast --no-exact --no-secondary --no-cpu-baseline"
run() {   # name, lib, waves
  if [ -n "$2" ]; then export GANDTR_HIP_LIB=$PWD/tmpbin/lib_$2.so; else unset GANDTR_HIP_LIB; fi
  GDT_C_WAVES=$3 timeout -k 10 300 $B > gpurun_out/ab_$1.log 2> gpurun_out/ab_$1.err
  python - "$1" <<'PY'
import json, sys
l = [x for x in open('gpurun_out/ab_%s.log' % sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[1], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); r = d['roofline']
print("%-14s %8.1f img/s  dominant %.4f ms  all: %s" % (sys.argv[1], d['value'], r['avg_launch_ms'], {k[:28]: v['ms_per_step'] for k, v in r['all_conv_kernels'].items()}))
PY
}
for rep in 1 2; do
  run w4_$rep "" 4
  run w8_$rep "" 8
  for v in "$@"; do run ${v}_w4_$rep $v 4; run ${v}_w8_$rep $v 8; done
done
