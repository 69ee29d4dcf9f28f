#!/bin/bash
# Developer script (GPU box): job indices drawn per wave in batches (draw_job) against one returned atomic per draw; whole frame, 8-way shard, the other configs.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/out.txt; tail -1 $O/pytest.log >> $O/out.txt
for b in 0 16 32 64; do
  echo "== bunny batch $b: $(ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 3 2>&1 | grep 'rep' | tail -2 | tr '\n' ' ')" >> $O/out.txt
done
for b in 0 32; do
  echo "== bunny plain batch $b: $(ORT_EXCHANGE=0 ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  for w in "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do set -- $w
    echo "== $1 batch $b: $(ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
for b in 0 32; do
  ORT_JOB_BATCH=$b PROXY_WORLDS=4,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/p_$b.json > $O/p_$b.log 2>&1
  echo "== proxy batch $b: $(grep '^N=' $O/p_$b.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
for t in 2 4 16; do
  ORT_BATCH_TAIL=$t PROXY_WORLDS=8 timeout -k 10 200 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/pt_$t.json > $O/pt_$t.log 2>&1
  echo "== proxy batch 32 tail $t: $(grep '^N=' $O/pt_$t.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
cat $O/out.txt
