"""Summarise the FETCH_SIZE / WRITE_SIZE passes of `rocprofv3 --pmc` over tools/scan_point.py c4 (6 calls of the
64-query scan over 1 M x 768 fp16) into profiles/pmc_scan.json: HBM-side bytes of ONE whole call.
usage: python3 tools/pmc_scan.py <fetch_dir> <write_dir> <out.json>
Units as tools/pmc_traffic.py: KB; read bytes = 2 * FETCH_SIZE * 1024 (gfx950 half-count of wide reads), WRITE_SIZE exact."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hair-centric-image-retrieval_amd"))


def load(d, counter):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter and re.search("sim_topk|merge|select|sim_scan", r["Kernel_Name"]):
            name = "scan" if "sim_topk_scan" in r["Kernel_Name"] else ("select" if "select" in r["Kernel_Name"] else "merge")
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    f, w = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    calls = 6
    rd = sum(2 * 1024 * sum(v) for v in f.values()) / calls
    wr = sum(1024 * sum(v) for v in w.values()) / calls
    alg = 1_000_000 * 768 * 2 + 64 * 768 * 2 + 64 * 16 * 12
    res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of tools/scan_point.py c4 (64 queries, 1 M x 768 "
                   "fp16, k = 16), all launches of a call summed, averaged over 6 calls. KB units; read bytes = "
                   "2*FETCH_SIZE*1024 (gfx950 half-count of wide reads), WRITE_SIZE exact.",
           "read_bytes_per_call": int(rd), "write_bytes_per_call": int(wr), "whole_call_bytes": int(rd + wr),
           "algorithmic_bytes": alg, "ratio": (rd + wr) / alg,
           "per_kernel_read_bytes": {k: int(2 * 1024 * sum(v) / calls) for k, v in f.items()}}
    try:
        from hcir._lib import build_id
        res["src_hash"] = build_id()   # identity of the binary the counters were taken on
    except Exception as e:  # noqa: BLE001
        res["src_hash"] = None
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
