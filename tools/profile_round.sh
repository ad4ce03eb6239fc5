#!/bin/bash
# The round's rocprofv3 evidence, run ON the GPU box from the repo root:  bash tools/profile_round.sh <outdir>
#   1. --kernel-trace --stats over the bench.py command, ONE step in flight (--no-pipeline: with two steps in flight on
#      two streams the launches overlap and every duration stretches; the roofline's own HIP-event measurement is
#      taken on one stream too)                                           -> <outdir>/kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of the same command   -> <outdir>/pmc_traffic.json (tools/pmc_traffic.py)
#   3. the same two passes over the 64-query streaming scan               -> <outdir>/pmc_scan.json    (tools/pmc_scan.py)
# Counter passes run alone with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section); the program follows
# `--` directly (no env / sh -c hop).  Copy the summaries into profiles/ afterwards.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/${1:-gpurun_out/prof}
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$O/stats" --output-format csv -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-pipeline > "$O/bench_stats.json" 2> "$O/stats.err"; echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/fetch" --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$O/fetch.err"; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$O/write" --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> "$O/write.err"; echo "write done"
rocprofv3 --pmc FETCH_SIZE -d "$O/sfetch" --output-format csv -- python3 "$R/tools/scan_point.py" c4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$O/swrite" --output-format csv -- python3 "$R/tools/scan_point.py" c4 > /dev/null 2>&1; echo "scan pmc done"
cd "$R"
python3 tools/pmc_traffic.py "$O/fetch" "$O/write" "$O/pmc_traffic.json" | tail -3
python3 tools/pmc_scan.py "$O/sfetch" "$O/swrite" "$O/pmc_scan.json" | tail -12
cp "$O"/stats/*/*kernel_stats.csv "$O/kernel_stats.csv"
rm -rf "$O/fetch" "$O/write" "$O/sfetch" "$O/swrite" "$O/stats"
ls -la "$O"
