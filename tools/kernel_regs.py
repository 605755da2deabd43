"""dev: condenses `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin) to one line per kernel: demangled name, VGPRs, AGPRs, spills, scratch, LDS"""
import re, subprocess, sys
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r"remark: .*?: (Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPR Spill|SGPR Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", l)
    if not m: m = re.search(r"remark: \[.*\]\s*$", l) and None
    m = re.search(r"(Function Name|  VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", l)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name": cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n") if rows else []
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n).replace("void ", "")
    if pat and not re.search(pat, n): continue
    print("%-46s vgpr %3s agpr %3s  vspill %3s sspill %3s scratch %4s  occ %s" % (n, r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
