"""dev: averages the [c stamp] lines of a stamp-variant run (stdin) per kernel variant"""
import sys, re, collections
acc = collections.defaultdict(list)
for l in sys.stdin:
    m = re.search(r'MODE (\d+) FORM (\d+) BN (\d+).*bodies (\d+), chunk barriers (\d+), tile barrier (\d+), epilogue (\d+) cycles; total per wave (\d+)', l)
    if m: acc[m.group(1, 2, 3)].append([int(x) for x in m.group(4, 5, 6, 7, 8)])
for k, v in sorted(acc.items()):
    n = len(v); print("MODE %s FORM %s BN %s" % k, "n=%d" % n, "bodies/cbar/tbar/epi per tile, total per wave:", [round(sum(x[i] for x in v) / n) for i in range(5)])
