#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c16
timeout -k 10 600 python3 -m pytest tests/test_step_gpu.py -x -q -k "zeroes or golden" > gpurun_out/r03c16/step.log 2>&1; echo "step rc=$?"; tail -3 gpurun_out/r03c16/step.log
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "ema_sgd or transpose" > gpurun_out/r03c16/kern.log 2>&1; echo "kern rc=$?"; tail -2 gpurun_out/r03c16/kern.log
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r03c16/bench_$tag.log 2>&1; tail -1 gpurun_out/r03c16/bench_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['ms_per_step_windows'], 'host', d['host_enqueue_ms_per_step'], d['host_enqueue_idle_queue_ms'])"; }
b zero0 S4F_FUSED_ZERO_GRAD=0
b zero1 S4F_FUSED_ZERO_GRAD=1
b zero0b S4F_FUSED_ZERO_GRAD=0
b zero1b S4F_FUSED_ZERO_GRAD=1
