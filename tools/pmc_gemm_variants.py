#!/usr/bin/env python3
"""Per-template HBM-side read bytes of the big GEMM kernel for several builds of the same sources.
usage: python3 tools/pmc_gemm_variants.py <out.json> tag=<rocprof dir of `--pmc FETCH_SIZE -- python3 tools/gemm_point.py`> ...
FETCH_SIZE is in KB and reports half of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): bytes = 2 * 1024 * v."""
import collections
import csv
import glob
import json
import re
import sys


def load(d):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    names = {r["Dispatch_Id"]: r["Kernel_Name"] for r in csv.DictReader(open(kt[0]))} if kt else {}
    grid = {}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != "FETCH_SIZE":
            continue
        name = names.get(r["Dispatch_Id"], r.get("Kernel_Name", "?"))
        m = re.search(r"(gemm_f16_\w+_kernel<\d+)", name)
        if m:
            acc[m.group(1) + ">"].append(float(r["Counter_Value"]) * 2 * 1024)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out = sys.argv[1]
    m = 880 * 197
    alg = {"0": ("qkv", m * 768 * 2 + 2304 * 768 * 2), "1": ("fc1", m * 768 * 2 + 3072 * 768 * 2),
           "6": ("proj/fc2 (avg)", (m * 768 * 2 + 768 * 768 * 2 + m * 768 * 2 + m * 3072 * 2 + 768 * 3072 * 2 + m * 768 * 2) / 2)}
    res = {}
    for spec in sys.argv[2:]:
        tag, _, d = spec.partition("=")
        res[tag] = {}
        for k, v in sorted(load(d).items()):
            e = re.search(r"<(\d+)>", k).group(1)
            name, a = alg.get(e, ("?", None))
            res[tag][k] = {"shape": name, "read_bytes_per_launch": v, "algorithmic_read_bytes": a,
                           "ratio": v / a if a else None}
            print(f"{tag:12s} {k:28s} {name:16s} {v / 1e9:7.3f} GB read, x{v / a:.2f} of algorithmic" if a else f"{tag} {k} {v}")
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
