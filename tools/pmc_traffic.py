"""Summarise the FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc` over bench.py into
profiles/<tag>_pmc_traffic.json (per kernel: average HBM-side bytes per launch).

usage: python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in
KB; FETCH_SIZE reports half of wide coalesced reads -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact.
"""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hair-centric-image-retrieval_amd"))


def load(d, counter):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    names = {}
    if kt:
        names = {r["Dispatch_Id"]: r["Kernel_Name"] for r in csv.DictReader(open(kt[0]))}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != counter:
            continue
        name = names.get(r["Dispatch_Id"], r.get("Kernel_Name", "?"))
        acc[short(name)].append(float(r["Counter_Value"]))
    return acc


def short(name):
    m = re.search(r"(gemm_f16_big_kernel<\d+)", name)
    if m:
        return m.group(1) + ">"
    m = re.search(r"(attn_fwd2_kernel|attn_fwd_kernel<\d+>|layernorm_f16_kernel|sim_topk_scan|topk_merge_kernel|patch_embed_kernel|"
                  r"ln_stats_finalize_kernel|topk_refine_kernel|gemm_f16_kernel<\d+)", name)
    if m:
        return m.group(1)
    m = re.search(r"sim_topk_scanI(\w+?)_?Li(\d+)", name)
    if m:
        return "sim_topk_scan"
    return None


def main():
    fd, wd, out = sys.argv[1:4]
    f, w = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(k for k in (set(f) | set(w)) if k is not None):
        fv, wv = f.get(k, []), w.get(k, [])
        n = max(len(fv), len(wv))
        fk = sum(fv) / max(len(fv), 1)
        wk = sum(wv) / max(len(wv), 1)
        kernels[k] = {"launches": n, "fetch_size_kb": round(fk, 1), "write_size_kb": round(wk, 1),
                      "read_bytes_corrected": int(2 * fk * 1024), "write_bytes": int(wk * 1024),
                      "bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
    big = {k: v for k, v in kernels.items() if k.startswith("gemm_f16_big_kernel")}
    tot_n = sum(v["launches"] for v in big.values())
    avg = sum(v["bytes_per_launch"] * v["launches"] for v in big.values()) / max(tot_n, 1)
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes of `python3 bench.py --steps 3 "
                   "--warmup 1 --no-cpu-baseline`. KB units; read bytes = 2*FETCH_SIZE*1024 (gfx950 half-count of wide "
                   "reads, MI355X_MICROARCH.md HBM section), WRITE_SIZE exact; L2->fabric requests, Infinity-Cache "
                   "hits included. gemm_f16_big_kernel<0> first qkv, <8> LayerNorm-folded qkv, <9> LayerNorm-folded "
                   "fc1+GELU, <7> proj / fc2 with row statistics, <6> plain fp16-residual epilogue.",
           "kernels": kernels, "gemm_f16_big_kernel_avg_bytes_per_launch": int(avg)}
    try:  # which build these counters belong to (bench.py reports them only for the same sources)
        from hcir._lib import build_id
        res["src_hash"] = build_id()   # identity of the binary the counters were taken on
    except Exception as e:  # noqa: BLE001
        res["src_hash"] = None
        print("source hash unavailable:", e)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v["bytes_per_launch"] for k, v in kernels.items()}, indent=1))
    print("gemm avg bytes/launch", int(avg))


if __name__ == "__main__":
    main()
