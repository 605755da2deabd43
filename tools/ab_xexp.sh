#!/bin/bash
# dev: GeM-ResNet-101 32 x 1024^2 with / without the fused 3x3 + expand launch (conv3x3_expand_rb.hip), per-op table + the bench's secondary line
O=gpurun_out
python tools/r101_ops.py > $O/xexp_ops_on.log 2>&1 && GDT_CONV_XEXP=0 python tools/r101_ops.py > $O/xexp_ops_off.log 2>&1 && \
python bench.py --steps 30 --no-cpu-baseline --no-fast --no-exact > $O/xexp_bench_on.json 2> $O/xexp_bench_on.err && \
GDT_CONV_XEXP=0 python bench.py --steps 30 --no-cpu-baseline --no-fast --no-exact > $O/xexp_bench_off.json 2> $O/xexp_bench_off.err
tail -3 $O/xexp_ops_on.log; tail -3 $O/xexp_ops_off.log
