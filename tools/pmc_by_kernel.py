"""dev: average rocprofv3 --pmc counters per kernel (substring filter)   usage: pmc_by_kernel.py <counter_collection.csv> <filter>"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if sys.argv[2] not in k:
        continue
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", k)
    e = acc[m.group(1) if m else k[:60]][row["Counter_Name"]]
    e[0] += 1; e[1] += float(row["Counter_Value"])
for k, v in acc.items():
    print(k, {c: round(x[1] / x[0] / 1e6, 3) for c, x in sorted(v.items())}, "launches", max(x[0] for x in v.values()))
