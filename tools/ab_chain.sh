#!/bin/bash
# dev: the chained layer3 launch (phase C) against the separate reduce launches, back to back inside one gpurun call
for v in 1 0 1 0; do
  GDT_XEXP_CHAIN=$v timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-fast --no-exact --no-cpu-baseline > gpurun_out/chain_$v.log 2> gpurun_out/chain_$v.err || { echo "bench failed (chain $v)"; tail -5 gpurun_out/chain_$v.err; exit 1; }
  python - $v <<'PY'
import json,sys
l=[x for x in open("gpurun_out/chain_%s.log"%sys.argv[1]) if x.startswith("{")]
d=json.loads(l[-1]); s=d["secondary"]; r=s["roofline"]
print("chain", sys.argv[1], "desc/s", s["value"], "ms", s["ms_per_step"], {k[:44]:v["ms_per_step"] for k,v in r["all_conv_kernels"].items() if "expand" in k or "1x1" in k})
PY
done
