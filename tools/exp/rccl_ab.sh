#!/bin/bash
# A/B of the N > 1 options through real RCCL (one-rank group): tools/exp/rccl_world1.py.  Run on the GPU box.
mkdir -p gpurun_out/r2c
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python tools/exp/rccl_world1.py -- --steps 20 --warmup 5 > gpurun_out/r2c/rccl_$name.txt 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/r2c/rccl_$name.txt; return 1; }
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
for line in open(f'gpurun_out/r2c/rccl_{name}.txt'):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(f'{name:28s} {d["ms_per_step"]:7.2f} ms/step  host {d["host_enqueue_ms_per_step"]:6.2f} ms  loss {d["losses"]["loss"]:.4f}')
    if line.startswith('[rccl_world1]'):
        print('   ', line.strip())
PY
}
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2c/rccl_plain.txt 2>&1 && python - <<'PY'
import json
for line in open('gpurun_out/r2c/rccl_plain.txt'):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(f'{"plain (no process group)":28s} {d["ms_per_step"]:7.2f} ms/step  host {d["host_enqueue_ms_per_step"]:6.2f} ms  loss {d["losses"]["loss"]:.4f}')
PY
run default A=1 &&
run aux_lockstep S4F_AUX_LOCKSTEP=1 &&
run aux_decode_lockstep S4F_AUX_LOCKSTEP=1 S4F_DECODE_LOCKSTEP=1 &&
run no_eager_sgd S4F_EAGER_SGD=0 &&
run lockstep_layout S4F_AUX_LOCKSTEP=1 S4F_DECODE_LOCKSTEP=1
