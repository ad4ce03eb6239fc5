#!/bin/bash
# Round-4 rocprofv3 evidence, run ON the GPU box from the repo root:  bash tools/profile_round4.sh <outdir>
# Everything tools/profile_round.sh records (kernel stats of the bench command, FETCH_SIZE / WRITE_SIZE passes of it and
# of the 64-query scan), plus: SQ-counter passes over the inference GEMM shapes (tools/gemm_point.py) and over the
# training step (tools/bench_train.py), kernel stats of the training step and of the PNG decoder (tools/bench_png.py).
# Counter passes run alone with --kernel-trace only; the program follows `--` directly.  Second argument "extras-only":
# skip the tools/profile_round.sh part (the two halves fit one 20-minute box call each).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/${1:-gpurun_out/prof4}
mkdir -p "$O"
[ "$2" = "extras-only" ] || bash "$R/tools/profile_round.sh" "${1:-gpurun_out/prof4}"
cd /tmp && export TMPDIR=/tmp
sq() {  # sq <name> <program...>: three SQ passes
  local name=$1; shift
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$O/${name}_a" --output-format csv -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace -d "$O/${name}_b" --output-format csv -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$O/${name}_c" --output-format csv -- "$@" > /dev/null 2>&1
  (cd "$R" && python3 tools/pmc_sq.py "$O/pmc_sq_${name}.json" "$O/${name}_a" "$O/${name}_b" "$O/${name}_c" > "$O/pmc_sq_${name}.txt")
  rm -rf "$O/${name}_a" "$O/${name}_b" "$O/${name}_c"
  echo "sq $name done"
}
sq gemm python3 "$R/tools/gemm_point.py" 880
rocprofv3 --kernel-trace --stats -d "$O/train_stats" --output-format csv -- python3 "$R/tools/bench_train.py" 1024 > "$O/bench_train.txt" 2>&1; echo "train stats done"
cp "$O"/train_stats/*/*kernel_stats.csv "$O/train_kernel_stats.csv"; rm -rf "$O/train_stats"
sq train python3 "$R/tools/bench_train.py" 1024
rocprofv3 --kernel-trace --stats -d "$O/png_stats" --output-format csv -- python3 "$R/tools/bench_png.py" 880 > "$O/bench_png.txt" 2>&1; echo "png stats done"
cp "$O"/png_stats/*/*kernel_stats.csv "$O/png_kernel_stats.csv"; rm -rf "$O/png_stats"
ls -la "$O"
