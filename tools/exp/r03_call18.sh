#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c18
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r03c18/bench_$tag.log 2>&1; tail -1 gpurun_out/r03c18/bench_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['ms_per_step_windows'])" || tail -3 gpurun_out/r03c18/bench_$tag.log; }
b q3 GPU_MAX_HW_QUEUES=3
b q3_cus64 GPU_MAX_HW_QUEUES=3 S4F_SIDE_CUS=64
b q3_cus96 GPU_MAX_HW_QUEUES=3 S4F_SIDE_CUS=96
b q2_cus64 GPU_MAX_HW_QUEUES=2 S4F_SIDE_CUS=64
b q3_cus64_noeager GPU_MAX_HW_QUEUES=3 S4F_SIDE_CUS=64 S4F_EAGER_STREAM=side
b q4_cus64_noeagersgd S4F_SIDE_CUS=64 S4F_EAGER_SGD=0
