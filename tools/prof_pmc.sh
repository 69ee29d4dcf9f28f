#!/bin/bash
# Developer script: PMC passes for the path-trace kernel (run on the GPU box via gpurun).
# Counters go in separate passes, with --kernel-trace only (no trace domains alongside --pmc).
set -o pipefail
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${PMC_ARGS:-c3_bunny_room 1920 1080 64 64 1}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/tools/prof_run.py $ARGS > $OUT/pass$i.log 2>&1 || echo "pass $i ($set) failed" >> $OUT/failed.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'pt_persistent' not in r['Kernel_Name']: continue
        acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for k,v in acc.items():
        g.write('%s mean_per_dispatch %.6g n %d\n'%(k,sum(v)/len(v),len(v)))
print(open(out+'/summary.txt').read())
PY
