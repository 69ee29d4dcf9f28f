#!/bin/bash
# Developer script (GPU box): the bench lines of the four workloads with their PMC stamps in place, and rocprofv3 kernel stats of the headline command.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3lines
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/progress.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/stats
timeout -k 10 300 python3 bench.py --scene c2_analytic --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout -k 10 300 python3 bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
timeout -k 10 300 python3 bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout -k 10 300 python3 bench.py --policy tile32 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_tile32.json 2> $O/bench_tile32.err
ORT_BENCH_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 2 --warmup 1 --spp 128 > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc $?" >> $O/progress.txt
cat $O/progress.txt
for f in $O/bench.json $O/bench_c*.json $O/bench_tile32.json $O/rehearsal.json; do python3 -c "import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f', round(d['value'],1), round(r['kernel_ms'],2), r['bound'], round(r['frac'],3), r.get('traffic'))"; done
