#!/bin/bash
# Developer script (GPU box): five-waves plain loop by default -- tests, scaling proxy against four waves, where the exchange should start.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3five
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
run() { tag=$1; worlds=$2; shift; shift
  env "$@" PROXY_WORLDS=$worlds $P $O/p_$tag.json > $O/p_$tag.log 2>&1
  echo "== $tag: $(grep '^N=' $O/p_$tag.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/summary.txt; }
run default 1,2,4,8
run w4 1,2,4,8 ORT_WAVES5=0
run plain_w5 1,2,4,8 ORT_EXCHANGE=0
run exch 2,4,8 ORT_EXCHANGE=1
echo "proxies done" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_EXCHANGE=0 STRESS_SEED0=9300 timeout -k 10 400 python3 tools/stress_parity.py 20 100 > $O/stress_w5.log 2>&1; echo "stress w5 rc $?" >> $O/progress.txt
for sc in "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do set -- $sc
  echo "== $1 default: $(python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/summary.txt
done
tail -3 $O/pytest.log | head -1; cat $O/progress.txt; cat $O/summary.txt; tail -1 $O/stress_w5.log
