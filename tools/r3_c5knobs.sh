#!/bin/bash
# Developer script (GPU box): 1M-triangle scene under batched job draws -- descend threshold above 16, the 4-wide tree again, leaf sizes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c5knobs
rm -rf $O; mkdir -p $O
cd $R
for v in "X=1" "ORT_DESCEND_BELOW=24" "ORT_DESCEND_BELOW=32" "ORT_DESCEND_BELOW=24 ORT_REFILL_BELOW=40" "ORT_WIDE=1" "ORT_LEAF_TRI=2" "ORT_LEAF_TRI=8" "ORT_JOB_BATCH=256 ORT_BATCH_TAIL=32"; do
  echo "== c5 $v: $(env $v timeout -k 10 200 python3 tools/prof_run.py c5:708 3840 2160 256 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
done
cat $O/out.txt
