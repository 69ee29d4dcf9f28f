#!/bin/bash
# Developer script (GPU box): the job space in eight contiguous ranges, one per XCD (experiment build, no stealing), against the default build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3xcd
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
ORT_LIB=$L/libort_xcd.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/out.txt; tail -1 $O/pytest.log >> $O/out.txt
for w in "c5:708 3840 2160 256" "c4_dwarf_room 3840 2160 512" "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "testscene 1920 1080 512"; do set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_xcd.so"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
cat $O/out.txt
