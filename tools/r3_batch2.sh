#!/bin/bash
# Developer script (GPU box): batch size sweep for draw_job.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch2
rm -rf $O; mkdir -p $O
cd $R
for b in 64 128 256 1024; do
  echo "== bunny batch $b: $(ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  echo "== bunny plain batch $b: $(ORT_EXCHANGE=0 ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  for w in "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do set -- $w
    echo "== $1 batch $b: $(ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
  ORT_JOB_BATCH=$b PROXY_WORLDS=2,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/p_$b.json > $O/p_$b.log 2>&1
  echo "== proxy batch $b: $(grep '^N=' $O/p_$b.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
cat $O/out.txt
