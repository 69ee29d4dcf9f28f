#!/bin/bash
# Developer script (GPU box): where the exchange still pays with batched job draws -- long launches (bunny room 4096 spp), shards of the dwarf room.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch4
rm -rf $O; mkdir -p $O
cd $R
for v in "ORT_EXCHANGE=1" "ORT_EXCHANGE=0 ORT_WAVES5=0"; do
  echo "== bunny 4096 spp batch 128 $v: $(env $v ORT_JOB_BATCH=128 timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 4096 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  echo "== bunny 256 spp batch 64 $v: $(env $v ORT_JOB_BATCH=64 timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 256 64 3 2>&1 | grep 'rep 2' | tail -1)" >> $O/out.txt
  env $v ORT_JOB_BATCH=64 PROXY_WORLDS=1,2,4,8 timeout -k 10 400 python3 tools/scaling_proxy.py c4_dwarf_room 3840 2160 512 64 $O/p.json > $O/p.log 2>&1
  echo "== dwarf proxy batch 64 $v: $(grep '^N=' $O/p.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
cat $O/out.txt
