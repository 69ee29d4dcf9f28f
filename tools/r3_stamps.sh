#!/bin/bash
# Developer script (GPU box): PMC passes (tools/prof_pmc.sh) for the four bench workloads -> gpurun_out/stamps/<tag>_summary.txt.
# Turn them into stamps afterwards, in the dev container: tools/make_pmc_stamp.py (the stamp carries the kernel-source hash).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stamps
rm -rf $O; mkdir -p $O
cd $R
for w in "c3 c3_bunny_room 1920 1080 1024 64" "c2 c2_analytic 1920 1080 1024 64" "c4 c4_dwarf_room 3840 2160 512 64" "c5 c5:708 3840 2160 256 64"; do
  set -- $w
  PMC_ARGS="$2 $3 $4 $5 $6 1" bash tools/prof_pmc.sh > $O/$1_pmc.log 2>&1
  cp gpurun_out/pmc/summary.txt $O/$1_summary.txt
  [ -f gpurun_out/pmc/failed.txt ] && cp gpurun_out/pmc/failed.txt $O/$1_failed.txt
  rm -rf gpurun_out/pmc
  echo "$1 done" >> $O/progress.txt
done
