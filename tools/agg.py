"""dev tool: aggregate a tools/dump_ops.py log by (kind, kernel variant)"""
import re, sys, collections
rows = [l.split() for l in open(sys.argv[1]) if re.match(r'^\s*\d+ (conv|inorm|maxpool|gem|input)', l)]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    key = (r[1], r[3]); agg[key][0] += 1; agg[key][1] += float(r[4]); agg[key][2] += float(r[7]) if len(r) > 7 else 0
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-8s %7s n=%3d ms=%7.3f gflop=%6.0f tflops=%5.0f" % (k[0], k[1], v[0], v[1], v[2], v[2] / v[1] if v[1] else 0))
print("total %.3f ms" % sum(v[1] for v in agg.values()))
