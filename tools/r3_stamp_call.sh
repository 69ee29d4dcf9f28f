#!/bin/bash
# Developer script (GPU box): GPU tests, then the PMC passes of the four bench workloads (tools/r3_stamps.sh).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3final2
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 STRESS_SEED0=9500 timeout -k 10 400 python3 tools/stress_parity.py 10 60 > $O/stress_production.log 2>&1; echo "stress rc $?" >> $O/progress.txt
PROXY_REPS=3 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/scaling_proxy.json > $O/scaling_proxy.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
ORT_EXCHANGE=1 timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_exchange.log 2>&1
ORT_EXCHANGE=0 timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_plain.log 2>&1
bash tools/r3_stamps.sh > $O/stamps.log 2>&1; cp -r gpurun_out/stamps $O/stamps; echo "stamps done" >> $O/progress.txt
tail -3 $O/pytest.log | head -1; cat $O/progress.txt; grep "^N=" $O/scaling_proxy.log; tail -1 $O/stress_production.log
