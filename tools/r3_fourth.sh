#!/bin/bash
# Developer script (GPU box), round 3, fourth call: LPT without the fence, prologue A/B, C5 counters.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/progress.txt
P="timeout -k 10 300 python3 tools/scaling_proxy.py c3_bunny_room 1920 1080 1024 64"
PROXY_WORLDS=1,2,4,8 $P $O/p_default.json > $O/p_default.log 2>&1; echo "proxy rc $?" >> $O/progress.txt
ORT_LPT=0 PROXY_WORLDS=1,8 $P $O/p_nolpt.json > $O/p_nolpt.log 2>&1
ORT_EXCHANGE=1 PROXY_WORLDS=8 $P $O/p_exch.json > $O/p_exch.log 2>&1
ORT_EXCHANGE=1 ORT_LPT=8 PROXY_WORLDS=8 $P $O/p_exch_lpt8.json > $O/p_exch_lpt8.log 2>&1
ORT_LPT=6 PROXY_WORLDS=8 $P $O/p_lpt6.json > $O/p_lpt6.log 2>&1
ORT_LPT=8 PROXY_WORLDS=8 $P $O/p_lpt8.json > $O/p_lpt8.log 2>&1
ORT_LPT=12 PROXY_WORLDS=8 $P $O/p_lpt12.json > $O/p_lpt12.log 2>&1
ORT_LIB=$R/offline_raytracer_amd/lib/libort_nopro.so PROXY_WORLDS=1,8 $P $O/p_nopro.json > $O/p_nopro.log 2>&1
echo "proxies done" >> $O/progress.txt
for sc in c2_analytic testscene glass_room c4_dwarf_room; do
  echo "== $sc pro:   $(python3 tools/prof_run.py $sc 1920 1080 256 64 2 | tail -1)" >> $O/ab_prologue.txt
  echo "== $sc nopro: $(ORT_LIB=$R/offline_raytracer_amd/lib/libort_nopro.so python3 tools/prof_run.py $sc 1920 1080 256 64 2 | tail -1)" >> $O/ab_prologue.txt
done
echo "ab done" >> $O/progress.txt
for v in "ORT_WIDE=0" "ORT_WIDE=1" "ORT_WIDE=0 ORT_LEAF_TRI=8" "ORT_WIDE=0 ORT_LEAF_TRI=2" "ORT_WIDE=1 ORT_REFILL_BELOW=24 ORT_DESCEND_BELOW=8" "ORT_WIDE=1 ORT_LEAF_TRI=8" "ORT_WIDE=0 ORT_LPT=0"; do
  echo "== c5 $v: $(env $v python3 tools/prof_run.py c5:708 3840 2160 128 64 2 | tail -1)" >> $O/c5_variants.txt
done
echo "c5 variants done" >> $O/progress.txt
ORT_WIDE=0 PMC_ARGS="c5:708 3840 2160 64 64 1" bash tools/prof_pmc.sh > $O/pmc_c5.log 2>&1; cp gpurun_out/pmc/summary.txt $O/pmc_c5_binary.txt
ORT_WIDE=0 PMC_ARGS="c5:708 3840 2160 64 64 1" bash tools/prof_pmc2.sh > $O/pmc2_c5.log 2>&1; cp gpurun_out/pmc2/summary.txt $O/pmc2_c5_binary.txt
echo "pmc done" >> $O/progress.txt
tail -3 $O/pytest.log; cat $O/progress.txt; for f in $O/p_*.log; do echo "== $f"; grep "^N=" $f; done; cat $O/ab_prologue.txt $O/c5_variants.txt
