#!/bin/bash
# Developer script (GPU box): exchange against plain loop (batched job draws) over mesh sizes: where is the crossover in the tree's SAH cost?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch5
rm -rf $O; mkdir -p $O
cd $R
for s in var_bunny_s3 var_bunny_s8 var_bunny_s12 var_dwarf_s0.008 var_dwarf_s0.012 var_dwarf_s0.03; do
  for v in "ORT_EXCHANGE=1" "ORT_EXCHANGE=0 ORT_WAVES5=0" "ORT_EXCHANGE=0 ORT_WAVES5=1"; do
    echo "== $s $v: $(env $v ORT_JOB_BATCH=64 timeout -k 10 200 python3 tools/prof_run.py $s 1920 1080 512 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
cat $O/out.txt
