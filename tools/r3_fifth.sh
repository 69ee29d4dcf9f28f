#!/bin/bash
# Developer script (GPU box), round 3, fifth call: how many chunks to issue sorted, at every shard size; exchange on/off under it.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3e
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
for l in 0 4 8 12 14; do
  ORT_LPT=$l PROXY_WORLDS=1,2,4,8 $P $O/p_lpt$l.json > $O/p_lpt$l.log 2>&1
done
echo "lpt sweep done" >> $O/progress.txt
for l in 8 14; do
  ORT_EXCHANGE=0 ORT_LPT=$l PROXY_WORLDS=1,2,4,8 $P $O/p_plain_lpt$l.json > $O/p_plain_lpt$l.log 2>&1
  ORT_EXCHANGE=1 ORT_LPT=$l PROXY_WORLDS=4,8 $P $O/p_exch_lpt$l.json > $O/p_exch_lpt$l.log 2>&1
done
echo "exchange sweep done" >> $O/progress.txt
for l in 0 1 4 6; do
  ORT_LPT=$l PROXY_WORLDS=1,8 timeout -k 10 300 python3 tools/scaling_proxy.py c4_dwarf_room 3840 2160 512 64 $O/p_c4_lpt$l.json > $O/p_c4_lpt$l.log 2>&1
done
echo "c4 done" >> $O/progress.txt
tail -3 $O/pytest.log; cat $O/progress.txt; for f in $O/p_*.log; do echo "== $f"; grep "^N=" $f; done
