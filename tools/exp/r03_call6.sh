#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c6; mkdir -p "$out"; cd $GRAFT_REPO_ROOT
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > "$out/bench_$tag.log" 2>&1; tail -1 "$out/bench_$tag.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
b base A=1
b wg_chain S4F_LAYER_WG_SIDE=0
b noside S4F_SIDE_STREAM=0
b base2 A=1
b wg_chain2 S4F_LAYER_WG_SIDE=0
