#!/bin/bash
# Developer script (GPU box): with batched job draws, which loop for the diffuse flavour on a cache-resident tree -- exchange, plain at four waves, plain at five.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch3
rm -rf $O; mkdir -p $O
cd $R
for b in 64 128; do
  for w in "c3_bunny_room 1920 1080 1024" "c4_dwarf_room 3840 2160 512"; do set -- $w
    echo "== $1 batch $b exchange: $(ORT_EXCHANGE=1 ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
    echo "== $1 batch $b plain w4: $(ORT_EXCHANGE=0 ORT_WAVES5=0 ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
    echo "== $1 batch $b plain w5: $(ORT_EXCHANGE=0 ORT_WAVES5=1 ORT_JOB_BATCH=$b timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
for v in "ORT_EXCHANGE=0 ORT_WAVES5=0" "ORT_EXCHANGE=0 ORT_WAVES5=1" "ORT_EXCHANGE=1"; do
  env $v ORT_JOB_BATCH=64 PROXY_WORLDS=1,2,4,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/p.json > $O/p.log 2>&1
  echo "== proxy batch 64 $v: $(grep '^N=' $O/p.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
for w in "c2_analytic 1920 1080 1024" "c5:708 3840 2160 256"; do set -- $w
  echo "== $1 batch 128 w4: $(ORT_WAVES5=0 ORT_JOB_BATCH=128 timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  echo "== $1 batch 128 exchange: $(ORT_EXCHANGE=1 ORT_JOB_BATCH=128 timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
done
cat $O/out.txt
