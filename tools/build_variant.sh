#!/bin/bash
# Developer script: build the working tree's library as offline_raytracer_amd/lib/libort_<tag>.so (A/B runs through ORT_LIB).
# usage: [W5FLAGS=...] tools/build_variant.sh <tag> [extra hipcc flags for ort_kernels.hip and the host sources, e.g. -DORT_PROLOGUE_DEFER=0]
# (W5FLAGS: extra flags for the five-waves unit ort_kernels_w5.hip, which is compiled on its own)
set -e
cd "$(dirname "$0")/../offline_raytracer_amd/csrc"
tag=$1; shift
F="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-math-errno --offload-arch=gfx950 -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt"
mkdir -p ../lib/.build
/opt/rocm/bin/hipcc $F $W5FLAGS -mllvm -disable-machine-licm -x hip -c ort_kernels_w5.hip -o ../lib/.build/ort_kernels_w5_$tag.o
/opt/rocm/bin/hipcc $F "$@" -shared -x hip ort_api.cpp ort_parse.cpp ort_tree.cpp ort_reftree.cpp ort_hdr.cpp ort_comm.cpp ort_kernels.hip \
  -x none ../lib/.build/ort_kernels_w5_$tag.o -o ../lib/libort_$tag.so -ldl
ls -la ../lib/libort_$tag.so
