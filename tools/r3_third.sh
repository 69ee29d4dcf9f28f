#!/bin/bash
# Developer script (GPU box), round 3, third call: tests (wide tree, sorted issue), N=8 proxy A/B, C5 wide vs binary, r2 lib vs now.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
PROXY_WORLDS=1,2,4,8 $P $O/p_default.json > $O/p_default.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
ORT_LPT=0 PROXY_WORLDS=1,8 $P $O/p_nolpt.json > $O/p_nolpt.log 2>&1
ORT_EXCHANGE=1 PROXY_WORLDS=4,8 $P $O/p_exch.json > $O/p_exch.log 2>&1
ORT_EXCHANGE=1 ORT_LPT=0 PROXY_WORLDS=8 $P $O/p_exch_nolpt.json > $O/p_exch_nolpt.log 2>&1
ORT_EXCHANGE=0 PROXY_WORLDS=1,8 $P $O/p_plain.json > $O/p_plain.log 2>&1
ORT_EXCHANGE=0 ORT_LPT=0 PROXY_WORLDS=1,8 $P $O/p_plain_nolpt.json > $O/p_plain_nolpt.log 2>&1
ORT_LPT=2 PROXY_WORLDS=8 $P $O/p_lpt2.json > $O/p_lpt2.log 2>&1
ORT_LPT=8 PROXY_WORLDS=8 $P $O/p_lpt8.json > $O/p_lpt8.log 2>&1
ORT_EXCHANGE=0 ORT_LIB=$R/offline_raytracer_amd/lib/libort_r2.so PROXY_WORLDS=1,8 $P $O/p_r2lib_plain.json > $O/p_r2lib_plain.log 2>&1
ORT_LIB=$R/offline_raytracer_amd/lib/libort_r2.so PROXY_WORLDS=1 $P $O/p_r2lib.json > $O/p_r2lib.log 2>&1
echo "proxies done" >> $O/progress.txt
for w in 0 1; do
  ORT_WIDE=$w timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_wide$w.json 2> $O/bench_c5_wide$w.err
done
ORT_LIB=$R/offline_raytracer_amd/lib/libort_r2.so timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_r2lib.json 2> $O/bench_c5_r2lib.err
echo "c5 done" >> $O/progress.txt
timeout -k 10 300 python3 bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
ORT_LPT=0 timeout -k 10 300 python3 bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_nolpt.json 2> $O/bench_c4_nolpt.err
timeout -k 10 300 python3 bench.py --scene c2_analytic --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
tail -3 $O/pytest.log; cat $O/progress.txt; for f in $O/p_*.log; do echo "== $f"; grep "^N=" $f; done
for f in $O/bench*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],1), round(d['roofline']['kernel_ms'],2))"; done
