#!/bin/bash
# Developer script (GPU box): environment-knob sweep.  usage: tools/r2_sweep.sh "<VAR=val,VAR=val ...> ..." [scene] [spp]
R=$GRAFT_REPO_ROOT
cd $R
SC=${2:-c3_bunny_room}
SPP=${3:-1024}
for combo in $1; do
  echo "== $combo $SC spp $SPP: $(env $(echo $combo | tr ',' ' ') python tools/prof_run.py $SC 1920 1080 $SPP 64 2 | tail -1)"
done
