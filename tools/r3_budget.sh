#!/bin/bash
# Developer script (GPU box): analytic-prologue budget re-swept under batched job draws (all-lobes scenes).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3budget
rm -rf $O; mkdir -p $O
cd $R
for b in 8 14 18 24 40; do
  for w in "c2_analytic 1920 1080 1024" "testscene 1920 1080 512" "glass_room 1920 1080 1024" "c3_bunny_room 1920 1080 1024"; do set -- $w
    echo "== $1 budget $b: $(ORT_ANALYTIC_PROLOGUE=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
cat $O/out.txt
