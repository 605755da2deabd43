"""dev: compact instruction trace + class counts of the loops of one kernel in a hipcc -S listing.
usage: tools/asm_trace.py <file.s> <kernel-name substring> [min loop length] [chars of trace]
  M fp16 MFMA   X MX MFMA   r ds_read   w ds_write   G global load   S global store   n s_nop   . VALU   , SALU   [..] s_waitcnt   | barrier"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
nchar = int(sys.argv[4]) if len(sys.argv) > 4 else 0
text = open(path).read().split("\n")
start = next(i for i, l in enumerate(text) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
end = next(i for i in range(start, len(text)) if text[i].strip().startswith(".Lfunc_end"))
lines = text[start:end]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] >= minlen:
        loops.append((labels[m.group(1)], i))
def sym(s):
    op = s.split()[0]
    if op.startswith("v_mfma_scale"): return "X"
    if op.startswith("v_mfma"): return "M"
    if op.startswith("s_waitcnt"): return "[" + s.split(None, 1)[1].replace("vmcnt", "vm").replace("lgkmcnt", "lg").replace(" ", "") + "]"
    if op.startswith("ds_read") or op.startswith("ds_bpermute"): return "r"
    if op.startswith("ds_write"): return "w"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "G"
    if op.startswith("global_store") or op.startswith("buffer_store"): return "S"
    if op.startswith("s_nop"): return "n"
    if op.startswith("s_barrier"): return "|"
    if op.startswith("v_"): return "."
    if op.startswith("s_"): return ","
    return "?"
for a, b in loops:
    c = collections.Counter(); tr = []
    for l in lines[a:b]:
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"): continue
        op = s.split()[0]
        k = ("MFMA " + op) if op.startswith("v_mfma") else op if re.match(r"(ds_|global_|buffer_|s_waitcnt|s_nop|s_barrier|v_readlane|v_writelane|v_accvgpr)", op) else "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else op
        c[k.split("(")[0]] += 1; tr.append(sym(s))
    print("loop lines %d..%d (%d)" % (a, b, b - a))
    print("   " + "  ".join("%s %d" % kv for kv in sorted(c.items(), key=lambda x: -x[1])))
    waits = [t for t in tr if t.startswith("[vm(") ]
    print("   vmcnt waits:", collections.Counter(re.match(r"\[vm\((\d+)\)", t).group(1) for t in waits).most_common())
    if nchar: print("".join(tr)[:nchar])

# ---- issue model: one wave per SIMD issues in order; an MFMA occupies the matrix pipe for 32 cycles and the issue port for 8, so the run of
# other instructions up to the next MFMA is free while it fits in 24 cycles.  Costs: 4 cycles per instruction, s_nop N = 4 * (N + 1)... (rough)
def exposed(a, b):
    run = 0; tot = 0; n_runs = 0; worst = []
    for l in lines[a:b]:
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"): continue
        op = s.split()[0]
        if op.startswith("v_mfma"):
            if run > 24: tot += run - 24; worst.append(run)
            run = 0; n_runs += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_barrier"): pass
        elif op.startswith("s_nop"): run += 4 * (int(s.split()[1]) + 1)
        else: run += 4
    return tot, n_runs, sorted(worst)[-10:]
for a, b in loops:
    t, n, w = exposed(a, b)
    print("loop %d..%d: %d MFMAs (%d pipe cycles), issue cycles not covered by an MFMA shadow ~%d; longest runs %s" % (a, b, n, 32 * n, t, w))
