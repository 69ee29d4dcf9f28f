#!/bin/bash
# Developer script: cache / latency counters of the 1M-triangle scene for two ORT_DESCEND_BELOW settings.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c5pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in 0 8; do
  i=0
  SETS=("TCC_HIT_sum TCC_MISS_sum" "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU")
    if [ -n "$C5_PMC_SET1" ]; then SETS=("$C5_PMC_SET1" "$C5_PMC_SET2" "$C5_PMC_SET3"); fi
    for set in "${SETS[@]}"; do
    [ -z "$set" ] && continue
    i=$((i+1))
    ORT_DESCEND_BELOW=$t timeout -k 10 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/d${t}_pass$i -- python3 $R/tools/prof_c5.py 708 32 32 1 > $OUT/d${t}_pass$i.log 2>&1 || echo "d$t pass $i failed" >> $OUT/failed.txt
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/c5pmc'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/d*_pass*/**/*counter_collection.csv', recursive=True)):
    tag=f.split('/c5pmc/')[1].split('_')[0]
    for r in csv.DictReader(open(f)):
        if 'pt_persistent<false' not in r['Kernel_Name']: continue
        acc.setdefault((tag,r['Counter_Name']),[]).append(float(r['Counter_Value']))
for k,v in acc.items(): print('%s %s mean_per_dispatch %.6g n %d'%(k[0],k[1],sum(v)/len(v),len(v)))
PY
