#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c7; mkdir -p "$out"; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > "$out/kern.log" 2>&1; echo "attn tests rc=$?"; tail -3 "$out/kern.log"
timeout -k 10 600 python3 -m pytest tests/test_step_gpu.py -x -q > "$out/step.log" 2>&1; echo "step rc=$?"; tail -3 "$out/step.log"
for t in 0 1; do echo "== S4F_ATTN_TAIL=$t"; S4F_ATTN_TAIL=$t python3 tools/attn_probe.py 2>&1 | tail -8; done
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > "$out/bench_$tag.log" 2>&1; tail -1 "$out/bench_$tag.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'])"; }
b tail0 S4F_ATTN_TAIL=0
b tail1 S4F_ATTN_TAIL=1
b tail0b S4F_ATTN_TAIL=0
b tail1b S4F_ATTN_TAIL=1
