"""Turn the rocprofv3 SQ counter passes of tools/gpu_pmc_mfma.sh into the tracked MFMA-utilisation file (profiles/rNN_pmc_mfma.json).

    python profiles/summarise_pmc_sq.py <dir with p1/, p2/, ... pass directories> <out.json> [kernel substring ...]

Every pass directory holds one `*counter_collection.csv` (rocprofv3 --kernel-trace --pmc <counters> --output-format csv).  Per kernel
variant (template arguments kept: MODE / FORM matter) the counters are averaged over its launches.  Derived figures:

  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)   share of the CU's four matrix pipes that is busy
                    (MFMA_BUSY counts cycles summed over SIMDs, BUSY_CU counts cycles per CU: MI355X_MICROARCH.md, cycle constants)
  wait_inst_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                    issue stalls (both in quad-cycles)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    if not m:
        return name[:60]
    return m.group(1) + (m.group(2) or "").replace(" ", "")


def main():
    root, out_path, filters = sys.argv[1], sys.argv[2], sys.argv[3:] or ["conv3x3_halo_c"]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if not any(s in k for s in filters):
                    continue
                e = acc[short(k)][row["Counter_Name"]]
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    out = {"method": "rocprofv3 --kernel-trace --pmc <two SQ counters per pass> -- python3 bench.py --steps 3 --warmup 1 "
                     "--no-cpu-baseline --no-secondary --no-fast --no-exact (tools/gpu_pmc_mfma.sh); per-launch averages",
           "kernels": {}}
    for k, cs in sorted(acc.items()):
        rec = {"launches": max(v[0] for v in cs.values())}
        for c, (n, s) in sorted(cs.items()):
            rec[c] = round(s / n, 1)
        g = rec.get
        if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
            rec["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES")), 4)
        if g("SQ_WAIT_INST_ANY") and g("SQ_WAVE_CYCLES"):
            rec["wait_inst_frac"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        out["kernels"][k] = rec
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", out_path, len(out["kernels"]), "kernel variants")


if __name__ == "__main__":
    main()
