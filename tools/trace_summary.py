"""Per-kernel durations of the LAST `ncalls`-th of a rocprofv3 --kernel-trace CSV (one hcir_sim_topk call = a few
launches): usage trace_summary.py <dir> [pattern] [last_n]"""
import csv, glob, re, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "sim_topk|merge|sim_scan|refine"
last = int(sys.argv[3]) if len(sys.argv) > 3 else 8
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if re.search(pat, r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{name:60s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} vgpr {r['VGPR_Count']}")
print(f"span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
