#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c14
timeout -k 10 900 python3 -m pytest tests/test_step_gpu.py -x -q > gpurun_out/r03c14/step.log 2>&1; echo "step rc=$?"; tail -3 gpurun_out/r03c14/step.log
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r03c14/bench_$tag.log 2>&1; tail -1 gpurun_out/r03c14/bench_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['ms_per_step_windows'], 'host', d['host_enqueue_ms_per_step'], d['host_enqueue_idle_queue_ms'], 'loss', d['losses']['loss'])"; }
b fused0 S4F_FUSED_LAUNCH=0
b fused1 S4F_FUSED_LAUNCH=1
b fused0b S4F_FUSED_LAUNCH=0
b fused1b S4F_FUSED_LAUNCH=1
