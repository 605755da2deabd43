"""Turn two rocprofv3 counter passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) into the per-kernel
HBM-traffic file bench.py reads for `roofline.traffic`.

    python profiles/summarise_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> "<workload note>"

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced
reads (MI355X_MICROARCH.md, HBM / rocprofv3 section), WRITE_SIZE is in KB as documented."""
import collections
import csv
import json
import re
import sys


def short(name):
    """kernel key as bench.py's kernel_name(): template kernels keep their tile arguments"""
    m = re.search(r"(conv1x1_rb_kernel|conv3x3_expand_rb_kernel|conv_bneck_kernel|conv_stem_pair_pool_kernel|conv3x3_halo_c16_kernel|conv3x3_halo_c_kernel|conv3x3_halo_rb_kernel|conv3x3_halo_kernel|conv3x3_halo_x3_kernel|conv_igemm_rb_kernel|conv_igemm_x3_kernel|conv_igemm_kernel|conv_head7_kernel|conv_stem_kernel)(<[^>]*>)?", name)
    if not m:
        g = re.search(r"(clahe_\w+_kernel|resample_\w+_kernel|reduce_kernel)(<[^>]*>)?", name)      # section-8f rows
        if g:
            return g.group(1) + (g.group(2) or "")
        return re.sub(r"^_ZN\d+_GLOBAL__N_\d+", "", name)[:48]
    base, args = m.group(1), m.group(2)
    if not args:
        return base
    a = [v.strip() for v in args[1:-1].split(",")]
    if base == "conv1x1_rb_kernel":
        return base
    if base == "conv3x3_expand_rb_kernel":       # <PH, CHAIN>: CHAIN = the next block's reduce conv as a third phase (round 5)
        return "%s<%s>%s" % (base, a[0], "[+next reduce]" if len(a) > 1 and a[1] == "true" else "")
    if base == "conv_igemm_kernel":
        return "%s<%s,%s>%s" % (base, a[0], a[1], "[norm]" if len(a) > 4 and a[4] == "true" else "")
    if base in ("conv3x3_halo_kernel", "conv3x3_halo_x3_kernel"):
        return "%s<%s>" % (base, a[1])
    if base == "conv3x3_halo_c16_kernel":        # <MODE>: the 256-column resblock conv on the 16 x 16 MFMA shapes
        return "%s<256>[mode %s]" % (base, a[0])
    if base == "conv3x3_halo_c_kernel":          # <BN, WGM, WGN, MODE, FORM>
        form = {"0": "", "1": "[transposed]", "2": "[stride-2]"}.get(a[4] if len(a) > 4 else "0", "")
        return "%s<%s>%s[mode %s]" % (base, a[0], form, a[3])
    if base == "conv3x3_halo_rb_kernel":
        return "%s<%s>%s[mode %s]" % (base, a[0], "[transposed]" if len(a) > 4 and a[4] == "true" else "", a[3])
    if base == "conv_stem_kernel":
        return "%s<%d taps>%s" % (base, int(a[0]) ** 2, "[f16c]" if len(a) > 2 and a[2] == "true" else "")
    if base == "conv_igemm_rb_kernel":
        return "%s<%s>%s" % (base, a[0], "[norm]" if a[3] == "true" else "")
    return "%s<%s>" % (base, a[0])


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                e = acc[short(row["Kernel_Name"])]
                e[0] += 1
                e[1] += float(row["Counter_Value"])
    return acc


def main():
    fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "",
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes; bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section). "
                     "Variants of one kernel template that bench.py reports under one name are averaged together.",
           "kernels": {}}
    merged = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
    for k, (n, v) in fetch.items():
        merged[k][0] += n; merged[k][1] += v
    for k, (n, v) in write.items():
        merged[k][2] += n; merged[k][3] += v
    for k, (nf, vf, nw, vw) in sorted(merged.items()):
        if not nf or not nw:
            continue
        out["kernels"][k] = {"launches": nf, "FETCH_SIZE_KB_avg": round(vf / nf, 1), "WRITE_SIZE_KB_avg": round(vw / nw, 1),
                             "hbm_bytes_per_launch_corrected": int((2 * vf / nf + vw / nw) * 1024)}
    # bench.py names: the dominant kernel is reported without the mode suffix -> add the launch-weighted union
    groups = collections.defaultdict(list)
    for k in out["kernels"]:
        groups[re.sub(r"\[(mode \d+|norm)\]$", "", k)].append(k)
    for base, ks in groups.items():
        if base in out["kernels"] and len(ks) == 1:
            continue
        n = sum(out["kernels"][k]["launches"] for k in ks)
        b = sum(out["kernels"][k]["hbm_bytes_per_launch_corrected"] * out["kernels"][k]["launches"] for k in ks) / n
        out["kernels"][base + " (all modes)" if base in out["kernels"] else base] = {
            "launches": n, "hbm_bytes_per_launch_corrected": int(b), "of": ks}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", sys.argv[3], len(out["kernels"]), "kernels")


if __name__ == "__main__":
    main()
