#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c17
b() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r03c17/bench_$tag.log 2>&1; tail -1 gpurun_out/r03c17/bench_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['ms_per_step'], d['ms_per_step_windows'], 'loss', round(d['losses']['loss'],4))" || tail -3 gpurun_out/r03c17/bench_$tag.log; }
b base A=1
b cus64 S4F_SIDE_CUS=64
b cus96 S4F_SIDE_CUS=96
b cus128 S4F_SIDE_CUS=128
b cus32 S4F_SIDE_CUS=32
b base2 A=1
b cus64b S4F_SIDE_CUS=64
