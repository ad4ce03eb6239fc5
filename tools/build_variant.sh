#!/bin/bash
# Build an A/B variant of libhcir.so: tools/build_variant.sh <tag> "<extra hipcc flags>" [file.hip ...]
# Recompiles the listed sources (default: gemm.hip) with the extra flags and links them with the in-tree objects
# of everything else into tools/_libhcir_<tag>.so (git-ignored; it travels to the GPU box with the snapshot).
set -e
tag=$1; flags=$2; shift 2
files=${@:-gemm.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/hair-centric-image-retrieval_amd/csrc
make -s -C "$src" -j8
tmp=/tmp/hcir_var_$tag; mkdir -p "$tmp"
objs=""
for o in "$src"/*.o; do
  b=$(basename "$o" .o)
  if echo " $files " | grep -q " $b.hip "; then
    extra=""; [ "$b" = attn_bwd ] && extra="-fno-slp-vectorize"   # as in csrc/Makefile
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Wno-inline-asm $extra $flags -c "$src/$b.hip" -o "$tmp/$b.o"
    objs="$objs $tmp/$b.o"
  else
    objs="$objs $o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$root/tools/_libhcir_$tag.so"
echo "built tools/_libhcir_$tag.so"
