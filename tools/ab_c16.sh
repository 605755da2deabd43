#!/bin/bash
# dev: conv3x3_halo_c16 (16 x 16 MFMA shapes) against conv3x3_halo_c inside one gpurun call: layer tests, bench A/B, stamps
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_f16c.py -x -q -m gpu -k "halo_c_conv3x3" 2>&1 | tail -3
B="python bench.py --steps 50 --warmup 10 --no-fast --no-exact --no-secondary --no-cpu-baseline"
run() {   # name, lib, c16 on/off
  if [ -n "$2" ]; then export GANDTR_HIP_LIB=$PWD/tmpbin/lib_$2.so; else unset GANDTR_HIP_LIB; fi
  GDT_CONV_HALO_C16=$3 timeout -k 10 300 $B > gpurun_out/ab_$1.log 2> gpurun_out/ab_$1.err
  python - "$1" <<'PY'
import json, sys
l = [x for x in open('gpurun_out/ab_%s.log' % sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[1], "FAILED"); sys.exit(0)
d = json.loads(l[-1]); r = d['roofline']; c = r['clocks_during_timed_region']
print("%-14s %8.1f img/s  dominant %.4f ms  %d MHz %d W  all: %s" % (sys.argv[1], d['value'], r['avg_launch_ms'], c['sclk_mhz_median'], c['power_w_mean'], {k[:24]: v['ms_per_step'] for k, v in r['all_conv_kernels'].items()}))
PY
}
for rep in 1 2; do
  run c16_$rep "" 1
  run c_$rep "" 0
  for v in "$@"; do run ${v}_$rep $v 1; done
done
unset GANDTR_HIP_LIB
if [ -f tmpbin/lib_stamp16.so ]; then GANDTR_HIP_LIB=$PWD/tmpbin/lib_stamp16.so timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-fast --no-exact --no-secondary --no-cpu-baseline 2>&1 >/dev/null | python tools/stamp_summary.py; fi
