#!/bin/bash
# Developer script: memory-pipeline PMC passes (TA / TCP / SQ VMEM levels) for the path-trace kernel.
# Counters go in separate passes, with --kernel-trace only (no trace domains alongside --pmc).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${PMC_ARGS:-c3_bunny_room 1920 1080 64 64 1}"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/tools/prof_run.py $ARGS > $OUT/pass$i.log 2>&1 || echo "pass $i ($set) failed" >> $OUT/failed.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc2'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'pt_persistent' not in r['Kernel_Name']: continue
        acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for k,v in acc.items():
        g.write('%s mean_per_dispatch %.6g n %d\n'%(k,sum(v)/len(v),len(v)))
print(open(out+'/summary.txt').read())
if os.path.exists(out+'/failed.txt'): print(open(out+'/failed.txt').read())
PY
