#!/bin/bash
# Developer script (GPU box): compiler flags that keep loop-invariant code (f64 polynomial constants, addresses) out of scalar registers held across the lane loop.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3licm
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
for w in "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "testscene 1920 1080 512" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do
  set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_nolicm.so" "ORT_LIB=$L/libort_sink.so"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/licm.txt
  done
done
for v in "X=1" "ORT_LIB=$L/libort_nolicm.so" "ORT_LIB=$L/libort_sink.so"; do
  echo "== c3 plain $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v ORT_EXCHANGE=0 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/licm.txt
done
cat $O/licm.txt
