#!/bin/bash
# Developer script (GPU box): every XCD draws from a job counter of its own over every eighth row of 8x8 blocks (experiment build, no stealing), against the default build.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3xcd
rm -rf $O; mkdir -p $O
cd $R
L=$R/offline_raytracer_amd/lib
cat > $O/img.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from offline_raytracer_amd import api
sc = api.Scene.load_scn(os.path.join(os.environ["GRAFT_REPO_ROOT"], "data", "c3_bunny_room.scn")).commit().upload(0)
img, st = sc.render(1000, 562, 128, 7, "chunk", chunk=16)
np.save(sys.argv[1], img)
PY
timeout -k 10 120 python3 $O/img.py $O/a.npy > /dev/null 2>&1
ORT_LIB=$L/libort_xcd.so timeout -k 10 120 python3 $O/img.py $O/b.npy > /dev/null 2>&1
python3 -c "import numpy as np; a=np.load('$O/a.npy'); b=np.load('$O/b.npy'); print('images bit-equal:', bool((a.view('<u4')==b.view('<u4')).all()))" >> $O/out.txt 2>&1
for w in "c5:708 3840 2160 256" "c4_dwarf_room 3840 2160 512" "c3_bunny_room 1920 1080 1024" "c2_analytic 1920 1080 1024" "testscene 1920 1080 512"; do set -- $w
  for v in "X=1" "ORT_LIB=$L/libort_xcd.so"; do
    echo "== $1 $(echo $v | sed 's#ORT_LIB=[^ ]*/libort_##'): $(env $v timeout -k 10 200 python3 tools/prof_run.py $1 $2 $3 $4 64 2 2>&1 | grep 'rep 1' | tail -1)" >> $O/out.txt
  done
done
cat $O/out.txt
