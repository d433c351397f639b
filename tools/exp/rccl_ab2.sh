#!/bin/bash
# round 4: what the N > 1 fixed cost (one-rank RCCL group) is made of after the group flush / tap node
mkdir -p gpurun_out/r2c
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python tools/exp/rccl_world1.py -- --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2c/rccl_$name.txt 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/r2c/rccl_$name.txt; return 1; }
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
for line in open(f'gpurun_out/r2c/rccl_{name}.txt'):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(f'{name:28s} {d["ms_per_step"]:7.2f} ms/step  host {d["host_enqueue_ms_per_step"]:6.2f} ms  loss {d["losses"]["loss"]:.4f}  collectives/step {d["config"].get("grad_collectives_per_step")}')
    if line.startswith('[rccl_world1]'):
        print('   ', line.strip())
PY
}
run default A=1 && run no_group_flush S4F_GROUP_FLUSH=0 && run no_tap_split S4F_TAP_SPLIT=0 && run neither S4F_GROUP_FLUSH=0 S4F_TAP_SPLIT=0 && run default_again A=1
