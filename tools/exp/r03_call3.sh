#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c3; mkdir -p "$out"; cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm or residual or patch_embed or gelu_resid or folded_tail or dense_modes" > "$out/kern.log" 2>&1; echo "kern rc=$?"; tail -3 "$out/kern.log"
timeout -k 10 900 python3 -m pytest tests/test_step_gpu.py -x -q > "$out/step.log" 2>&1; echo "step rc=$?"; tail -5 "$out/step.log"
for r in fp32 auto; do
  S4F_RESID=$r python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_$r.log" 2>&1
  tail -1 "$out/bench_$r.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$r', d['ms_per_step'], d['losses'])"
done
