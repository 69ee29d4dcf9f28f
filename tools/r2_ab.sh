#!/bin/bash
# Developer script (GPU box): A/B kernel builds.  usage: tools/r2_ab.sh "<lib suffixes, '-' = default>" [scenes] [spp]
R=$GRAFT_REPO_ROOT
cd $R
SC=${2:-"c3_bunny_room c2_analytic"}
SPP=${3:-256}
for v in $1; do
  if [ "$v" = "-" ]; then L=$R/offline_raytracer_amd/lib/libort.so; else L=$R/offline_raytracer_amd/lib/libort_$v.so; fi
  for sc in $SC; do
    echo "== lib $v scene $sc spp $SPP: $(ORT_LIB=$L python tools/prof_run.py $sc 1920 1080 $SPP 64 2 | tail -1)"
  done
done
