#!/bin/bash
# Developer script (GPU box): 5 and 6 waves per SIMD on the all-lobes scenes and the 1M-triangle scene; is machine LICM off needed?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3w56
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
for w in "c2_analytic 1920 1080 1024" "c5:708 3840 2160 256" "testscene 1920 1080 512" "glass_room 1920 1080 512" "c4_dwarf_room 3840 2160 512"; do
  set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_w5.so ORT_BLOCKS_PER_CU=5" "ORT_LIB=$L/libort_w5l.so ORT_BLOCKS_PER_CU=5" "ORT_LIB=$L/libort_w6.so ORT_BLOCKS_PER_CU=6 ORT_EXCHANGE=0"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w56.txt
  done
done
echo "== c4 plain w5: $(ORT_LIB=$L/libort_w5.so ORT_BLOCKS_PER_CU=5 ORT_EXCHANGE=0 python3 tools/prof_run.py c4_dwarf_room 3840 2160 512 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w56.txt
echo "== c4 plain w4: $(ORT_EXCHANGE=0 python3 tools/prof_run.py c4_dwarf_room 3840 2160 512 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/w56.txt
cat $O/w56.txt
