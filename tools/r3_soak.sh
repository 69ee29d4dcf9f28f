#!/bin/bash
# Developer script (GPU box): parity hunt on the final build of the round -- HIP path against the oracle, bit for bit, every kernel family forced in turn.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3soak
rm -rf $O; mkdir -p $O
cd $R
STRESS_COUNTERS=0 STRESS_SEED0=$((11000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 40 200 > $O/stress_default.log 2>&1; echo "default rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_WAVES5=1 ORT_EXCHANGE=0 STRESS_SEED0=$((12000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 30 150 > $O/stress_five_waves.log 2>&1; echo "five waves rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_WAVES5=0 ORT_EXCHANGE=0 STRESS_SEED0=$((13000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 30 120 > $O/stress_four_waves_plain.log 2>&1; echo "four waves plain rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_EXCHANGE=1 STRESS_SEED0=$((14000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 30 150 > $O/stress_exchange.log 2>&1; echo "exchange rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_DEBUG_FORCE_FALLBACK=0x3f STRESS_SEED0=$((15000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 20 100 > $O/stress_forced_recasts.log 2>&1; echo "forced recasts rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_LPT=0 STRESS_SEED0=$((16000+${SOAK_OFFSET:-0})) timeout -k 10 400 python3 tools/stress_parity.py 10 60 > $O/stress_chunk_major.log 2>&1; echo "chunk-major rc $?" >> $O/progress.txt
cat $O/progress.txt; for f in $O/stress_*.log; do echo "$(basename $f): $(grep -c ': ok' $f) ok, $(grep -c DIFF $f) DIFF, paths $(grep ': ok' $f | awk '{s+=$(NF-5)} END {print s}')"; done
