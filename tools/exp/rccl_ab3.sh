#!/bin/bash
# round 4: which switch carries the N > 1 fixed cost (one-rank RCCL group), all on one box
mkdir -p gpurun_out/r2c
show() { python3 - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d = json.loads(l); print(f'{sys.argv[2]:44s} {d["ms_per_step"]:7.2f} ms/step  host {d["host_enqueue_ms_per_step"]:6.2f}  collectives {d["config"].get("grad_collectives_per_step")}', flush=True)
PY
}
run() { name=$1; shift; env "$@" timeout -k 10 300 python3 tools/exp/rccl_world1.py -- --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r2c/x_$name.txt 2>&1; show gpurun_out/r2c/x_$name.txt "$name"; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r2c/x_plain.txt 2>&1; show gpurun_out/r2c/x_plain.txt "plain (no process group)"
run defaults A=1
run no_lockstep S4F_AUX_LOCKSTEP=0 S4F_DECODE_LOCKSTEP=0
run lockstep_no_tap_split S4F_TAP_SPLIT=0
run aux_lockstep_only S4F_AUX_LOCKSTEP=1 S4F_DECODE_LOCKSTEP=0
run decode_lockstep_only S4F_AUX_LOCKSTEP=0 S4F_DECODE_LOCKSTEP=1
run no_lockstep_again S4F_AUX_LOCKSTEP=0 S4F_DECODE_LOCKSTEP=0
