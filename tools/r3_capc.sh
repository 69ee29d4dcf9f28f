#!/bin/bash
# Developer script (GPU box): stash capacities as compile-time constants (libort.so) against run-time values (libort_capv.so).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3capc
rm -rf $O; mkdir -p $O
cd $R
for rep in 1 2; do
for v in "X=1" "ORT_LIB=$R/offline_raytracer_amd/lib/libort_capv.so"; do
  echo "== c3 $(echo $v | sed 's#ORT_LIB=[^ ]*#capv#'): $(env $v python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 3 2>&1 | grep 'rep' | tail -2 | tr '\n' ' ')" >> $O/capc.txt
done
done
for v in "X=1" "ORT_LIB=$R/offline_raytracer_amd/lib/libort_capv.so"; do
  echo "== c4 $(echo $v | sed 's#ORT_LIB=[^ ]*#capv#'): $(env $v python3 tools/prof_run.py c4_dwarf_room 3840 2160 512 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/capc.txt
done
cat $O/capc.txt
