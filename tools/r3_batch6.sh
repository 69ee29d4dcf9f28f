#!/bin/bash
# Developer script (GPU box): batched job draws as the default -- suite, order inside a block ([chunk][pixel] / [pixel][chunk]), scaling proxy.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3batch6
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/out.txt; tail -1 $O/pytest.log >> $O/out.txt
for l in 1 2; do
  for w in "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "c4_dwarf_room 3840 2160 512" "c5:708 3840 2160 256" "testscene 1920 1080 512" "glass_room 1920 1080 1024"; do set -- $w
    echo "== $1 order $l: $(ORT_LPT=$l timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
  ORT_LPT=$l PROXY_WORLDS=1,2,4,8 timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64 $O/p_$l.json > $O/p_$l.log 2>&1
  echo "== proxy order $l: $(grep '^N=' $O/p_$l.log | sed 's/ mean.*->//; s/max //' | tr '\n' '|')" >> $O/out.txt
done
cat $O/out.txt
