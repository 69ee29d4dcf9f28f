#!/bin/bash
# Developer script (GPU box): s_setprio around the traversal loop (traversal above shading, shading above traversal) against the default build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prio
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
for w in "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do
  set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_pt1.so" "ORT_LIB=$L/libort_ps1.so"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/prio.txt
  done
done
cat $O/prio.txt
