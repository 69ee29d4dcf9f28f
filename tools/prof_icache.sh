#!/bin/bash
# Developer script: instruction-cache counters of the path-trace kernel for one scene (PMC_SCENE).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S=${PMC_SCENE:-testscene}
timeout -k 10 100 rocprofv3 --kernel-trace --pmc SQC_ICACHE_MISSES SQC_ICACHE_REQ SQC_TC_INST_REQ SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/tools/prof_run.py $S 1920 1080 64 64 1 > $OUT/p1.log 2>&1
python3 - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/icache'
for f in sorted(glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'pt_persistent' in r['Kernel_Name']: print(r['Kernel_Name'][:50], r['Counter_Name'], r['Counter_Value'])
PY
