"""Summarise `rocprofv3 --pmc <SQ counters>` passes over the training step (tools/bench_train.py) into a per-kernel
table: matrix-pipe busy, LDS busy, bank conflicts, wait fractions.

usage: python3 tools/pmc_sq.py <out.json> <pass_dir> [<pass_dir> ...]   |   <out.json> <earlier_summary.json>
Each pass directory holds one rocprofv3 run (counter_collection.csv [+ kernel_trace.csv]).  SQ_* cycle counters are
summed over the chip's SQs in quad-cycle units for the *_CYCLES family (MI355X_MICROARCH.md, counters section); only
RATIOS of counters from the same pass are reported besides the raw per-launch averages."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hair-centric-image-retrieval_amd"))

KERNELS = [("gemm_f16_big_kernel<0", r"gemm_f16_big_kernel<0,"), ("gemm_f16_big_kernel<1", r"gemm_f16_big_kernel<1,"),
           ("gemm_f16_big_kernel<10", r"gemm_f16_big_kernel<10,"),
           ("gemm_f16_big_kernel<6", r"gemm_f16_big_kernel<6,"), ("gemm_f16_tn_kernel", r"gemm_f16_tn_kernel"),
           ("attn_bwd2_kernel", r"attn_bwd2_kernel"), ("attn_bwd_kernel", r"attn_bwd_kernel<"),
           ("attn_fwd_kernel", r"attn_fwd_kernel"), ("layernorm_bwd_kernel", r"layernorm_bwd_kernel"),
           ("gelu_bwd_colsum_kernel", r"gelu_bwd_colsum_kernel")]


def short(name):
    for tag, pat in KERNELS:
        if re.search(pat, name):
            return tag
    return None


def load(d):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    names, dur = {}, {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        k = short(names.get(r["Dispatch_Id"], r.get("Kernel_Name", "?")))
        if k is None:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] in dur:
            acc[k]["_us"].append(dur[r["Dispatch_Id"]])
    return acc


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    prior = None
    if dirs and dirs[0].endswith(".json"):   # re-derive the fractions of an earlier summary (raw averages kept in it)
        prior = json.load(open(dirs[0]))
        for k, row in prior["kernels"].items():
            res[k] = {n: v for n, v in row.items() if "frac" not in n and n != "shader_clock_ghz"}
        dirs = []
    for d in dirs:
        for k, counters in load(d).items():
            # the largest launches of a kernel (the block-sized ones): the upper half by counter magnitude
            for c, vals in counters.items():
                vals = sorted(vals)[len(vals) // 2:]
                res[k][c if c != "_us" else "avg_us_under_pmc"] = sum(vals) / len(vals)
    table = {}
    for k, c in res.items():
        row = {n: round(v, 1) for n, v in c.items()}
        g = c.get
        # GRBM_GUI_ACTIVE is reported SUMMED over the 8 XCDs (a 2.86 ms launch reads 47.3 M = 8 x 2.06 GHz x 2.86 ms);
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs in cycles (checked: attn_bwd2 = 32 cycles x its MFMA count)
        cyc = g("GRBM_GUI_ACTIVE") / 8 if g("GRBM_GUI_ACTIVE") else None
        if cyc and g("SQ_VALU_MFMA_BUSY_CYCLES"):
            row["mfma_busy_frac_of_simd_cycles"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc), 4)
        if cyc and g("SQ_LDS_IDX_ACTIVE"):
            row["lds_busy_frac_of_cu_cycles"] = round(g("SQ_LDS_IDX_ACTIVE") / (256 * cyc), 4)
        if cyc and c.get("avg_us_under_pmc"):
            row["shader_clock_ghz"] = round(cyc / c["avg_us_under_pmc"] / 1e3, 3)
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"):
            row["wait_any_frac_of_wave_cycles"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY"):
            row["issue_stall_frac_of_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY"):
            row["issuing_frac_of_wave_cycles"] = round(g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_IDX_ACTIVE") and g("SQ_LDS_BANK_CONFLICT") is not None:
            row["bank_conflict_frac_of_lds_cycles"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
        table[k] = row
    doc = {"note": "rocprofv3 --pmc passes (<= 4 SQ counters each, --kernel-trace only) over tools/bench_train.py; per "
                   "kernel the average over the larger half of its launches; fractions are ratios of counters of ONE pass",
           "kernels": table}
    try:
        from hcir._lib import build_id
        doc["src_hash"] = prior["src_hash"] if prior else build_id()
    except Exception as e:  # noqa: BLE001
        doc["src_hash"] = None
        print("build id unavailable:", e)
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
