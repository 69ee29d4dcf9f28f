#!/bin/bash
# Developer script: A/B of tree-builder knobs on the GPU box
for scene in c3_bunny_room c2_analytic testscene; do
for cfg in "0 2 4" "1 2 4" "1 4 4" "1 8 4" "1 8 2" "1 8 8"; do
  set -- $cfg
  echo -n "$scene split=$1 leaf_other=$2 leaf_tri=$3: "
  ORT_TREE_SPLIT_KINDS=$1 ORT_LEAF_OTHER=$2 ORT_LEAF_TRI=$3 python3 tools/prof_run.py $scene 1920 1080 64 64 2 | tail -1
done; done
