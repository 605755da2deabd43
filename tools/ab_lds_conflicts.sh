#!/bin/bash
# dev: LDS bank-conflict share of the shift forms under timing-only ablations (1 no halo staging, 2 no MX products / fp4 fragment reads, 8 no fp16 MFMAs)
R=$(pwd)
for b in base cabl1 cabl2 cabl8; do
  if [ $b != base ]; then export GANDTR_HIP_LIB=$R/tmpbin/lib_$b.so; fi
  bash tools/gpu_pmc_mfma.sh lds_$b tools/pmc_groups_lds.txt > /dev/null 2>&1
  echo "== $b"
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/pmc_mfma_lds_$b/pmc_mfma.json"))
for k,v in d["kernels"].items():
    if "halo_c_kernel" in k: print("  ", k, v.get("lds_conflict_frac"))
PY
done
