#!/bin/bash
# Developer script (GPU box), round 3, second call: tests incl. the new parity tests, LPT on/off through the scaling proxy, phase shares of c2 / c5.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_lpt.json > $O/proxy_lpt.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
ORT_LPT=0 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_nolpt.json > $O/proxy_nolpt.log 2>&1
ORT_EXCHANGE=1 PROXY_WORLDS=1,4,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_lpt_exch.json > $O/proxy_lpt_exch.log 2>&1
ORT_EXCHANGE=0 PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/proxy_lpt_plain.json > $O/proxy_lpt_plain.log 2>&1
PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c4_dwarf_room 3840 2160 512 64 $O/proxy_c4.json > $O/proxy_c4.log 2>&1
ORT_LPT=0 PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c4_dwarf_room 3840 2160 512 64 $O/proxy_c4_nolpt.json > $O/proxy_c4_nolpt.log 2>&1
echo "proxies done" >> $O/progress.txt
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
timeout -k 10 300 python3 tools/util_run.py c2_analytic 1920 1080 256 64 > $O/util_c2.log 2>&1
timeout -k 10 300 python3 tools/util_run.py c5:708 3840 2160 64 64 > $O/util_c5.log 2>&1
timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 256 64 > $O/util_c3_plain.log 2>&1
echo "util done" >> $O/progress.txt
tail -3 $O/pytest.log; cat $O/progress.txt; grep -h "^N=" $O/proxy_*.log
