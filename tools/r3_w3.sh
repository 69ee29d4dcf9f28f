#!/bin/bash
# Developer script (GPU box): 3 waves per SIMD (168 VGPRs, no spills) against 4 (128 VGPRs), per flavour and loop.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w3
rm -rf $O; mkdir -p $O
cd $R
W3="ORT_LIB=$R/offline_raytracer_amd/lib/libort_w3.so ORT_BLOCKS_PER_CU=3"
for sc in c2_analytic testscene c3_bunny_room; do
  for v in "ORT_EXCHANGE=0" "ORT_EXCHANGE=1" "$W3 ORT_EXCHANGE=0" "$W3 ORT_EXCHANGE=1"; do
    echo "== $sc $(echo $v | sed 's#ORT_LIB=[^ ]*#w3#'): $(env $v python3 tools/prof_run.py $sc 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w3.txt
  done
done
cat $O/w3.txt
