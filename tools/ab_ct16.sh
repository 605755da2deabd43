#!/bin/bash
# dev: conv_ct_c16 / conv_s2_c16 against the 32 x 32 shift forms inside one gpurun call
timeout -k 10 300 python -m pytest tests/test_hip_f16c.py -x -q -m gpu -k "transposed or stride2 or ragged" 2>&1 | tail -3
for v in "1 1" "0 0" "1 1" "0 0"; do set -- $v
  GDT_CONV_CT_C16=$1 GDT_CONV_S2_C16=$2 timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-fast --no-exact --no-secondary --no-cpu-baseline > gpurun_out/ct16_$1.log 2>&1
  python - $1 <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ct16_%s.log"%sys.argv[1]) if x.startswith("{")]
if not l: print("FAILED"); sys.exit(0)
d=json.loads(l[-1]); r=d["roofline"]
print(sys.argv[1], d["value"], r["clocks_during_timed_region"]["sclk_mhz_median"], {k[:30]:v["ms_per_step"] for k,v in r["all_conv_kernels"].items()})
PY
done
