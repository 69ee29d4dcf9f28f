#!/bin/bash
# Developer script (GPU box): parity tests, then kernel throughput of the headline scene and friends.
# usage: tools/r2_quick.sh [tag] [notest]
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-run}
mkdir -p $R/gpurun_out/r2
cd $R
if [ "$2" != "notest" ]; then
  timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r2/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/r2/${TAG}_tests.log; exit 1; }
  tail -2 gpurun_out/r2/${TAG}_tests.log
fi
{
for sc in c3_bunny_room c2_analytic testscene c4_dwarf_room glass_room; do
  echo "== $sc 1920x1080 64spp chunk64"; python tools/prof_run.py $sc 1920 1080 64 64 3
done
echo "== c3 1024spp chunk64"; python tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2
} > gpurun_out/r2/${TAG}_perf.log 2>&1
cat gpurun_out/r2/${TAG}_perf.log
