#!/bin/bash
# Developer script (GPU box): 5 waves per SIMD (96 VGPRs, 20 LDS stack entries, machine LICM off) against the default 4.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w5
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
for w in "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "c5:708 3840 2160 256"; do
  set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_w5.so ORT_BLOCKS_PER_CU=5" "ORT_LIB=$L/libort_w5.so ORT_BLOCKS_PER_CU=4"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w5.txt
  done
done
echo "== c3 plain w5: $(ORT_LIB=$L/libort_w5.so ORT_BLOCKS_PER_CU=5 ORT_EXCHANGE=0 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w5.txt
cat $O/w5.txt
