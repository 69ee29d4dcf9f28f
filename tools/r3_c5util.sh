#!/bin/bash
# Developer script (GPU box): lane-utilisation probes and phase shares of the 1M-triangle scene and the analytic scene (counters build, four waves).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c5util
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python3 tools/util_run.py c5:708 3840 2160 64 64 > $O/c5.log 2>&1
timeout -k 10 300 python3 tools/util_run.py c2_analytic 1920 1080 256 64 > $O/c2.log 2>&1
grep -h "^util\|^phase\|^c5\|^c2" $O/c5.log $O/c2.log
