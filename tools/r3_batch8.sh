#!/bin/bash
# Developer script (GPU box): plain-loop thresholds re-swept under batched job draws (whole frame and 8-way shard; other configs).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch8
rm -rf $O; mkdir -p $O
cd $R
P="timeout -k 10 200 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
for v in "X=1" "ORT_REFILL_BELOW=24" "ORT_REFILL_BELOW=32" "ORT_DESCEND_BELOW=12" "ORT_REFILL_BELOW=24 ORT_DESCEND_BELOW=12" "ORT_REFILL_BELOW=32 ORT_DESCEND_BELOW=12"; do
  env $v PROXY_WORLDS=1,8 $P $O/p.json > $O/p.log 2>&1
  echo "== bunny $v: $(grep '^N=' $O/p.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
for w in "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256" "testscene 1920 1080 512"; do set -- $w
  for v in "X=1" "ORT_REFILL_BELOW=24" "ORT_REFILL_BELOW=48" "ORT_DESCEND_BELOW=12" "ORT_DESCEND_BELOW=4"; do
    echo "== $1 $v: $(env $v timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
cat $O/out.txt
