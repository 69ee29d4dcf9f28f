#!/bin/bash
# Developer script (GPU box), round 3, seventh call: block-major issue vs cost-sorted issue, drain profile, other configs.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3g
rm -rf $O; mkdir -p $O
cd $R
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
run() { # tag, env...
  tag=$1; shift
  env "$@" ORT_DEBUG_FALLBACK=1 PROXY_WORLDS=1,8 $P $O/p_$tag.json > $O/p_$tag.log 2>&1
  echo "== $tag: $(grep '^N=' $O/p_$tag.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|') $(grep -h 'sorted issue' $O/p_$tag.log | sed 's/.*lists/lists/' | sort | uniq -c | tr '\n' ';')" >> $O/summary.txt
}
run lpt0 ORT_LPT=0
run lpt14 ORT_LPT=14
run lpt14_nosort ORT_LPT=14 ORT_LPT_SORT=0
run lpt12 ORT_LPT=12
run lpt12_nosort ORT_LPT=12 ORT_LPT_SORT=0
run lpt10 ORT_LPT=10
run lpt10_nosort ORT_LPT=10 ORT_LPT_SORT=0
run lpt8 ORT_LPT=8
run lpt8_nosort ORT_LPT=8 ORT_LPT_SORT=0
run lpt14_pm ORT_LPT=14 ORT_LPT_PIXEL_MAJOR=1
run lpt12_pm ORT_LPT=12 ORT_LPT_PIXEL_MAJOR=1
echo "sweep done" >> $O/progress.txt
ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=8 $P $O/p_drain.json > $O/p_drain.log 2>&1
ORT_LPT=0 ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=8 $P $O/p_drain_lpt0.json > $O/p_drain_lpt0.log 2>&1
ORT_LPT=10 ORT_DEBUG_DRAIN=1 PROXY_REPS=1 PROXY_WORLDS=8 $P $O/p_drain_lpt10.json > $O/p_drain_lpt10.log 2>&1
echo "drain done" >> $O/progress.txt
for v in "ORT_LPT=0" "ORT_LPT=2" "ORT_LPT=1" "ORT_LPT=2 ORT_LPT_SORT=0"; do
  echo "== c5 $v: $(env $v ORT_DEBUG_FALLBACK=1 python3 tools/prof_run.py c5:708 3840 2160 256 64 2 2>&1 | grep 'rep 1\|sorted issue' | tail -2 | tr '\n' ' ')" >> $O/others.txt
done
for v in "ORT_LPT=0" "ORT_LPT=6" "ORT_LPT=4" "ORT_LPT=6 ORT_LPT_SORT=0"; do
  echo "== c4 $v: $(env $v ORT_DEBUG_FALLBACK=1 python3 tools/prof_run.py c4_dwarf_room 3840 2160 512 64 2 2>&1 | grep 'rep 1\|sorted issue' | tail -2 | tr '\n' ' ')" >> $O/others.txt
done
for v in "ORT_LPT=0" "ORT_LPT=14" "ORT_LPT=10" "ORT_LPT=14 ORT_LPT_SORT=0"; do
  echo "== c2 $v: $(env $v ORT_DEBUG_FALLBACK=1 python3 tools/prof_run.py c2_analytic 1920 1080 1024 64 2 2>&1 | grep 'rep 1\|sorted issue' | tail -2 | tr '\n' ' ')" >> $O/others.txt
done
echo "others done" >> $O/progress.txt
cat $O/summary.txt; grep -h "drain:" $O/p_drain.log | head -3; echo; grep -h "drain:" $O/p_drain_lpt0.log | head -3; echo; grep -h "drain:" $O/p_drain_lpt10.log | head -3; cat $O/others.txt
