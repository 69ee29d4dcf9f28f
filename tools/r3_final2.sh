#!/bin/bash
# Developer script (GPU box): after the last kernel-file change of the round -- tests, bench lines, kernel stats, PMC passes per workload.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3final2
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
STRESS_COUNTERS=0 ORT_EXCHANGE=1 STRESS_SEED0=9100 timeout -k 10 400 python3 tools/stress_parity.py 10 60 > $O/stress_exchange.log 2>&1; echo "stress exchange rc $?" >> $O/progress.txt
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/stats
timeout -k 10 300 python3 bench.py --scene c2_analytic --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
ORT_EXCHANGE=0 timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_plain.json 2> $O/bench_c5_plain.err
echo "benches done" >> $O/progress.txt
PROXY_REPS=3 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/scaling_proxy.json > $O/scaling_proxy.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_exchange.log 2>&1
ORT_EXCHANGE=0 timeout -k 10 300 python3 tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_plain.log 2>&1
bash tools/r3_stamps.sh > $O/stamps.log 2>&1; cp -r gpurun_out/stamps $O/stamps; echo "stamps done" >> $O/progress.txt
tail -3 $O/pytest.log | head -1; cat $O/progress.txt; grep "^N=" $O/scaling_proxy.log
for f in $O/bench.json $O/bench_c*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['roofline']['kernel_ms'],2), d['roofline']['bound'], round(d['roofline']['frac'],3))"; done
