#!/bin/bash
# Developer script (GPU box): the 1M-triangle scene, round-2 library against the current one (4K, 256 spp in 64-sample jobs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c5
rm -rf $O; mkdir -p $O
cd $R
for v in "ORT_LPT=1" "ORT_LPT=0" "ORT_LIB=$R/offline_raytracer_amd/lib/libort_r2.so" "ORT_LPT=0 ORT_REFILL_BELOW=32 ORT_DESCEND_BELOW=16" "ORT_LPT=1 ORT_EXCHANGE=1" "ORT_LIB=$R/offline_raytracer_amd/lib/libort_nopro.so ORT_LPT=0"; do
  echo "== c5 $v: $(env $v ORT_DEBUG_FALLBACK=1 python3 tools/prof_run.py c5:708 3840 2160 256 64 2 2>&1 | grep 'rep 1\|issue order' | tail -2 | tr '\n' ' ')" >> $O/c5.txt
done
for v in "ORT_EXCHANGE=0" "ORT_EXCHANGE=1"; do
  echo "== c2 $v: $(env $v python3 tools/prof_run.py c2_analytic 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/c5.txt
  echo "== testscene $v: $(env $v python3 tools/prof_run.py testscene 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/c5.txt
done
cat $O/c5.txt
