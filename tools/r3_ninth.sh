#!/bin/bash
# Developer script (GPU box), round 3, ninth call: plain block-major issue; early endgame of the ray exchange (LDS flag).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3i
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
run x_e0 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=0
run x_e4 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=4
run x_e8 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=8
run x_e16 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=16
run x_e32 1,2,4,8 ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=32
run plain 1,2,4,8 ORT_EXCHANGE=0
run chunkmajor 1,8 ORT_LPT=0
echo "sweep done" >> $O/progress.txt
ORT_EXCHANGE=1 ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=1,8 $P $O/p_drain_x.json > $O/p_drain_x.log 2>&1
ORT_EXCHANGE=1 ORT_ENDGAME_JOBS=0 ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=1,8 $P $O/p_drain_x0.json > $O/p_drain_x0.log 2>&1
echo "drain done" >> $O/progress.txt
cat $O/summary.txt; grep -h "drain:" $O/p_drain_x.log | head -3; echo; grep -h "drain:" $O/p_drain_x0.log | head -3
