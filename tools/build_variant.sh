#!/bin/bash
# Developer script: build the working tree's library as offline_raytracer_amd/lib/libort_<tag>.so (A/B runs through ORT_LIB).
# usage: tools/build_variant.sh <tag> [extra hipcc flags, e.g. -DORT_LDS_STACK=16]
set -e
cd "$(dirname "$0")/../offline_raytracer_amd/csrc"
tag=$1; shift
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-math-errno --offload-arch=gfx950 -Wall -Wno-unused-function \
  -fhip-fp32-correctly-rounded-divide-sqrt "$@" -shared -x hip ort_api.cpp ort_parse.cpp ort_tree.cpp ort_reftree.cpp ort_hdr.cpp ort_comm.cpp ort_kernels.hip \
  -o ../lib/libort_$tag.so -ldl
ls -la ../lib/libort_$tag.so
