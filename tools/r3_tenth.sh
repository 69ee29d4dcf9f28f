#!/bin/bash
# Developer script (GPU box), round 3, tenth call: knob sweeps under the block-major issue (plain loop on an 8-way shard, exchange on the whole frame and on shards).
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3j
rm -rf $O; mkdir -p $O
cd $R
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
run() { # tag, worlds, env...
  tag=$1; worlds=$2; shift; shift
  env "$@" PROXY_WORLDS=$worlds $P $O/p_$tag.json > $O/p_$tag.log 2>&1
  echo "== $tag: $(grep '^N=' $O/p_$tag.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/summary.txt
}
run base 1,8
for r in 8 16 24 32; do run plain8_refill$r 8 ORT_EXCHANGE=0 ORT_REFILL_BELOW=$r; done
for d in 4 12 16; do run plain8_descend$d 8 ORT_EXCHANGE=0 ORT_DESCEND_BELOW=$d; done
echo "plain sweep done" >> $O/progress.txt
for c in 16 32 48; do run x8_cap$c 4,8 ORT_EXCHANGE=1 ORT_LONG_MIN=$c ORT_INFLIGHT_CAP=$c; done
run x8_cap32_e16 8 ORT_EXCHANGE=1 ORT_LONG_MIN=32 ORT_INFLIGHT_CAP=32 ORT_ENDGAME_JOBS=16
run x8_cap32_e2 8 ORT_EXCHANGE=1 ORT_LONG_MIN=32 ORT_INFLIGHT_CAP=32 ORT_ENDGAME_JOBS=2
echo "x8 sweep done" >> $O/progress.txt
for m in 48 96; do run x1_long$m 1 ORT_LONG_MIN=$m ORT_INFLIGHT_CAP=$m; done
for m in 16 48; do run x1_lrefill$m 1 ORT_LONG_REFILL=$m; done
for m in 12 20 24; do run x1_refill$m 1 ORT_REFILL_BELOW=$m; done
for m in 4 12; do run x1_descend$m 1 ORT_DESCEND_BELOW=$m; done
run x1_cap96_long64 1 ORT_INFLIGHT_CAP=96
run x1_cap128_long64 1 ORT_INFLIGHT_CAP=128
echo "x1 sweep done" >> $O/progress.txt
cat $O/summary.txt
