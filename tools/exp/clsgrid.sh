#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 512 768 1024 1536 2048 4096; do echo "cap $g"; S4F_CLS_STATS_GRID=$g timeout -k 10 100 python3 tools/exp/clsfuse_probe.py 2>/dev/null | grep "fused stats" | sed 's/.*| fused/   fused/'; done
