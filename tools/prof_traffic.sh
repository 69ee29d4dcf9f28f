#!/bin/bash
# Developer script: HBM traffic of the headline launch (1920x1080, 1024 spp) -- FETCH_SIZE and
# WRITE_SIZE in separate passes, plus a kernel trace for the matching duration.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="c3_bunny_room 1920 1080 1024 64 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- python3 $R/tools/prof_run.py $ARGS > $OUT/pass$i.log 2>&1 || echo "pass $i failed" >> $OUT/failed.txt
done
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/traffic'
acc=collections.OrderedDict()
for f in sorted(glob.glob(out+'/pass*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        acc.setdefault((r['Kernel_Name'][:40],r['Counter_Name']),[]).append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as g:
    for k,v in acc.items():
        g.write('%s | %s mean_per_dispatch %.6g n %d\n'%(k[0],k[1],sum(v)/len(v),len(v)))
print(open(out+'/summary.txt').read())
PY
