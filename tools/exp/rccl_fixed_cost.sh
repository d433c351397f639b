#!/bin/bash
# round 4: the N > 1 fixed cost on one box - bench.py plain, through a one-rank RCCL group (defaults), plain again
mkdir -p gpurun_out/r2c
show() { python3 - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); print(f'{sys.argv[2]:34s} {d["ms_per_step"]:7.2f} ms/step {d["ms_per_step_windows"]}  gradient collectives / step {d["config"].get("grad_collectives_per_step")}')
    if l.startswith('[rccl_world1]'):
        print('   ', l.strip()[:420])
PY
}
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/plain.txt 2>&1; show gpurun_out/r2c/plain.txt "plain (no process group)"
timeout -k 10 300 python3 tools/exp/rccl_world1.py -- --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/d1.txt 2>&1; show gpurun_out/r2c/d1.txt "one-rank RCCL group, defaults"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/plain2.txt 2>&1; show gpurun_out/r2c/plain2.txt "plain (no process group)"
