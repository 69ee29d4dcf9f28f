#!/bin/bash
# Developer script (GPU box): the ray exchange's thresholds re-swept on the scene that still uses it (dwarf room 4K) under batched job draws.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3xknobs
rm -rf $O; mkdir -p $O
cd $R
for v in "X=1" "ORT_LONG_MIN=48" "ORT_LONG_MIN=96" "ORT_LONG_REFILL=16" "ORT_LONG_REFILL=48" "ORT_INFLIGHT_CAP=96" "ORT_INFLIGHT_CAP=128" "ORT_PARK_MIN=4" "ORT_PARK_MIN=8" "ORT_ENDGAME_JOBS=8" "ORT_ENDGAME_JOBS=32" "ORT_DESCEND_BELOW=12" "ORT_REFILL_BELOW=32"; do
  echo "== dwarf $v: $(env $v timeout -k 10 200 python3 tools/prof_run.py c4_dwarf_room 3840 2160 512 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
done
cat $O/out.txt
