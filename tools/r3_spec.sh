#!/bin/bash
# Developer script (GPU box): speculative descent (a lane's first leaf of a round is put aside while it goes on descending) against the default build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3spec
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
ORT_LIB=$L/libort_spec.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/spec.txt; tail -2 $O/pytest.log >> $O/spec.txt
for w in "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256"; do
  set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_spec.so"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/spec.txt
  done
done
echo "== c3 plain default: $(ORT_EXCHANGE=0 timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/spec.txt
echo "== c3 plain spec: $(ORT_LIB=$L/libort_spec.so ORT_EXCHANGE=0 timeout -k 10 200 python3 tools/prof_run.py c3_bunny_room 1920 1080 1024 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/spec.txt
cat $O/spec.txt
