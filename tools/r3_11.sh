#!/bin/bash
# Developer script (GPU box), round 3, call 11: the exchange drains its stashes before the job space is empty.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
run() { # tag, worlds, env...
  tag=$1; worlds=$2; shift; shift
  env "$@" PROXY_WORLDS=$worlds $P $O/p_$tag.json > $O/p_$tag.log 2>&1
  echo "== $tag: $(grep '^N=' $O/p_$tag.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/summary.txt
}
run default 1,2,4,8
for e in 0 2 4 8 12 16; do run x_e$e 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=$e; done
run plain 1,2,4,8 ORT_EXCHANGE=0
echo "sweep done" >> $O/progress.txt
ORT_EXCHANGE=1 ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=8 $P $O/p_drain_x.json > $O/p_drain_x.log 2>&1
cat $O/summary.txt; grep -h "drain:" $O/p_drain_x.log | head -3; tail -3 $O/pytest.log | head -1
