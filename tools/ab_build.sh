#!/bin/bash
# A/B builds of one HIP source: tools/ab_build.sh <tag> <source.hip> [-D...]  ->  tools/_ab/libchambers_hip_<tag>.so
# (the other objects come from the in-tree build).  Benchmarks pick one up through CHB_AB_LIB (tools/gemm_bench.py, attn_bench.py).
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p tools/_ab
extra=""
case $src in augment.hip|imageio.hip|elementwise.hip) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -munsafe-fp-atomics $extra "$@" \
    -c chambers_amd/csrc/$src -o tools/_ab/${src%.hip}_$tag.o
objs=""
for f in augment imageio gemm layernorm attention attention_general elementwise metric vit_block; do
    if [ "$f.hip" == "$src" ]; then objs="$objs tools/_ab/${f}_$tag.o"; else objs="$objs chambers_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_ab/libchambers_hip_$tag.so $objs
echo tools/_ab/libchambers_hip_$tag.so
