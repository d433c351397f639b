#!/bin/bash
# round 5: bf16 output stores of the ping-pong GEMM with sc1 (the line leaves the XCD's L2 behind the store, so a tile's 128 KB of
# output stops evicting the operand panels its neighbours re-read): isolated launches of both builds, then the step interleaved
cd $GRAFT_REPO_ROOT
other=s4former_amd/libs4f_sc1.so
for lib in "" "S4F_LIB=$GRAFT_REPO_ROOT/$other" "" "S4F_LIB=$GRAFT_REPO_ROOT/$other"; do
  echo "== [$lib]"; env A=1 $lib timeout -k 10 200 python3 tools/exp/layer_gemm_time.py 16400 0 2>&1 | grep -v amdgpu.ids
done
tools/exp/ab_lib.sh $other
