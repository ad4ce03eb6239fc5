#!/bin/bash
# On the GPU box, from the repo root:  bash tools/pmc_gemm_variants.sh <outdir> tag[=lib] ...
# For every build variant: rocprofv3 --pmc FETCH_SIZE over tools/gemm_point.py (counter pass alone, with --kernel-trace
# only), then the per-template summary (tools/pmc_gemm_variants.py) and an interleaved timing A/B (tools/ab_gemm.py).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/$1; shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
specs=""; abs=""
for v in "$@"; do
  tag=${v%%=*}; lib=${v#*=}; [ "$lib" = "$v" ] && lib=""
  if [ -n "$lib" ]; then export HCIR_LIB_PATH=$R/$lib; else unset HCIR_LIB_PATH; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$O/$tag" --output-format csv -- python3 "$R/tools/gemm_point.py" 880 > "$O/$tag.log" 2>&1
  echo "pmc $tag done"
  specs="$specs $tag=$O/$tag"
  abs="$abs $tag=${lib:+$R/$lib}"
done
unset HCIR_LIB_PATH
cd "$R"
python3 tools/pmc_gemm_variants.py "$O/pmc_gemm_variants.json" $specs | tee "$O/pmc_gemm_variants.txt"
python3 tools/ab_gemm.py 880 $abs | tee "$O/ab_gemm.txt"
for v in "$@"; do tag=${v%%=*}; rm -rf "$O/$tag"; done
