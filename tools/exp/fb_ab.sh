#!/bin/bash
# same-box timing of the one-sweep attention backward of two library builds: tools/exp/fb_ab.sh <other .so>
cd $GRAFT_REPO_ROOT
for cfg in "16 1025 1" "16 1025 0" "8 1025 0" "8 2305 1" "8 2305 0"; do
  set -- $cfg
  for i in 1 2; do
    FB_B=$1 FB_N=$2 FB_BIAS=$3 python3 tools/exp/fb_time.py
    FB_B=$1 FB_N=$2 FB_BIAS=$3 S4F_LIB=$GRAFT_REPO_ROOT/${OTHER:-s4former_amd/libs4f_fbpost.so} python3 tools/exp/fb_time.py
  done
done
