#!/bin/bash
# Developer script (GPU box): counters of the 1M-triangle scene at 3840x2160, 256 spp in 64-sample jobs, for the deep-tree
# settings (default) against the cache-resident settings (descend early exit, early refill, analytic prologue).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/c5pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPP=${1:-256}
for tag in prev new; do
  if [ $tag = prev ]; then export ORT_LIB=$R/offline_raytracer_amd/lib/libort_prev.so; else unset ORT_LIB; fi
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${tag}_pass$i -- python3 $R/tools/prof_c5.py 708 $SPP 64 1 > $OUT/${tag}_pass$i.log 2>&1 || echo "$tag pass $i failed" >> $OUT/failed.txt
    echo "$tag pass $i done" >> $OUT/progress.txt
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/c5pmc'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/*_pass*/**/*counter_collection.csv', recursive=True)):
    tag=f.split('/c5pmc/')[1].split('_')[0]
    for r in csv.DictReader(open(f)):
        if 'pt_persistent<false' not in r['Kernel_Name']: continue
        acc.setdefault((r['Counter_Name'],tag),[]).append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for k,v in sorted(acc.items()): g.write('%s %s max_dispatch %.6g n %d\n'%(k[0],k[1],max(v),len(v)))
print(open(out+'/summary.txt').read())
PY
