#!/bin/bash
# Developer script (GPU box), round 3, sixth call: how a launch drains (sorted issue on / off), all configs under the new defaults.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3f
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
ORT_DEBUG_DRAIN=1 ORT_DEBUG_FALLBACK=1 PROXY_REPS=1 PROXY_WORLDS=1,8 $P $O/p_drain.json > $O/p_drain.log 2>&1
ORT_LPT=0 ORT_DEBUG_DRAIN=1 ORT_DEBUG_FALLBACK=1 PROXY_REPS=1 PROXY_WORLDS=1,8 $P $O/p_drain_nolpt.json > $O/p_drain_nolpt.log 2>&1
PROXY_WORLDS=1,2,4,8 $P $O/p_default.json > $O/p_default.log 2>&1
echo "drain done" >> $O/progress.txt
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
timeout -k 10 300 python3 bench.py --scene c2_analytic --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
ORT_LPT=0 timeout -k 10 300 python3 bench.py --scene c2_analytic --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c2_nolpt.json 2> $O/bench_c2_nolpt.err
timeout -k 10 300 python3 bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo "benches done" >> $O/progress.txt
timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3.log 2>&1
ORT_EXCHANGE=0 timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 256 64 > $O/util_c3_plain.log 2>&1
for b in 14 18 22; do
  echo "== c2 prologue budget $b: $(ORT_ANALYTIC_PROLOGUE=$b python3 tools/prof_run.py c2_analytic 1920 1080 256 64 2 | tail -1)" >> $O/budget.txt
  echo "== testscene prologue budget $b: $(ORT_ANALYTIC_PROLOGUE=$b python3 tools/prof_run.py testscene 1920 1080 256 64 2 | tail -1)" >> $O/budget.txt
done
tail -3 $O/pytest.log; cat $O/progress.txt; grep -h "^N=\|drain:\|sorted issue" $O/p_drain.log | head -30; echo; grep -h "^N=\|drain:\|sorted issue" $O/p_drain_nolpt.log | head -30; grep "^N=" $O/p_default.log; cat $O/budget.txt
for f in $O/bench*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['roofline']['kernel_ms'],2))"; done
