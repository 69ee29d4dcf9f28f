#!/bin/bash
# Developer script (GPU box): 8-way shard of the headline frame under batched job draws -- batch size, where batches stop, how the launch drains.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch7
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/out.txt; tail -1 $O/pytest.log >> $O/out.txt
P="timeout -k 10 200 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
for v in "X=1" "ORT_BATCH_TAIL=2" "ORT_BATCH_TAIL=4" "ORT_BATCH_TAIL=12" "ORT_JOB_BATCH=32 ORT_BATCH_TAIL=4" "ORT_JOB_BATCH=32 ORT_BATCH_TAIL=8" "ORT_JOB_BATCH=16 ORT_BATCH_TAIL=2" "ORT_REFILL_BELOW=8" "ORT_REFILL_BELOW=24" "ORT_REFILL_BELOW=32" "ORT_DESCEND_BELOW=4" "ORT_DESCEND_BELOW=12"; do
  env $v PROXY_WORLDS=8 $P $O/p.json > $O/p.log 2>&1
  echo "== N=8 $v: $(grep '^N=' $O/p.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=8 $P $O/pd.json > $O/pd.log 2>&1; grep -h "drain:" $O/pd.log | head -3 >> $O/out.txt
cat $O/out.txt
