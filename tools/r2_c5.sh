#!/bin/bash
# Developer script (GPU box): the 1M-triangle scene under knob combinations.  usage: tools/r2_c5.sh "<combo> ..." [spp]
R=$GRAFT_REPO_ROOT
cd $R
SPP=${2:-32}
for combo in $1; do
  echo "== $combo spp $SPP: $(env $(echo $combo | tr ',' ' ') python tools/prof_c5.py 708 $SPP $(( SPP < 64 ? SPP : 64 )) 2 2>&1 | grep "rep 1\|Error\|error" | tail -1)"
done
