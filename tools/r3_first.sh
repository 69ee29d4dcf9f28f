#!/bin/bash
# Developer script (GPU box), round 3, first call: GPU tests, the N=2 rehearsal from a plain process, scaling proxy, bench line.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
ORT_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --spp 128 > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc $?" >> $O/progress.txt
timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_default.json > $O/proxy_default.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
ORT_EXCHANGE=1 PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_exch.json > $O/proxy_exch.log 2>&1
ORT_EXCHANGE=0 PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_plain.json > $O/proxy_plain.log 2>&1
PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 32 $O/proxy_chunk32.json > $O/proxy_chunk32.log 2>&1
PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 16 $O/proxy_chunk16.json > $O/proxy_chunk16.log 2>&1
echo "proxies done" >> $O/progress.txt
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
tail -3 $O/pytest.log; cat $O/progress.txt; cat $O/rehearsal.json | cut -c1-400; tail -2 $O/proxy_*.log | cut -c1-300
