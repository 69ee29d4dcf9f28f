#!/bin/bash
# Developer script (GPU box): the measurements that go into profiles/ -- bench line, rocprofv3 kernel stats of the same
# command, PMC passes of the headline launch, secondary bench lines.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final
rm -rf $O; mkdir -p $O
cd $R
python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
echo "bench done" >> $O/progress.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
echo "stats done" >> $O/progress.txt
PMC_ARGS="c3_bunny_room 1920 1080 1024 64 1" bash tools/prof_pmc.sh > $O/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt $O/pmc_summary_headline.txt
echo "pmc done" >> $O/progress.txt
python bench.py --scene c2_analytic --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
python bench.py --scene c4_dwarf_room --width 3840 --height 2160 --spp 512 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err
echo "c2 c4 done" >> $O/progress.txt
python bench.py --scene c5_heightfield_708 --width 3840 --height 2160 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
echo "c5 done" >> $O/progress.txt
python bench.py --policy tile32 --spp 64 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_tile32.json 2> $O/bench_tile32.err
ORT_DEBUG_UTIL=1 python tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_1024.log 2>&1
ORT_EXCHANGE=0 ORT_DEBUG_UTIL=1 python tools/util_run.py c3_bunny_room 1920 1080 1024 64 > $O/util_c3_1024_plain.log 2>&1
cat $O/bench.json; cat $O/bench_kernel_stats.csv | head -5; cat $O/pmc_summary_headline.txt | head -30; for f in c2 c4 c5 tile32; do python3 -c "import json,sys; d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], d['roofline']['frac'], d['roofline']['frac_divergent'])"; done
