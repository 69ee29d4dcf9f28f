#!/bin/bash
# Developer script (GPU box): what a job draw costs -- the headline frame in 64- / 128- / 256-sample jobs at 4096 spp (same path count per run; tail = one job).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3chunklen
rm -rf $O; mkdir -p $O
cd $R
for ch in 64 128 256; do
  echo "== bunny 4096 spp chunk $ch: $(timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 4096 $ch 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  echo "== bunny 4096 spp chunk $ch plain: $(ORT_EXCHANGE=0 timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 4096 $ch 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
done
cat $O/out.txt
