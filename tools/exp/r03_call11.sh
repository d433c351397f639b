#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c11
timeout -k 10 900 python3 -m pytest tests/test_zz_dist_gpu.py -x -q > gpurun_out/r03c11/dist_tests.log 2>&1; echo "dist tests rc=$?"; tail -3 gpurun_out/r03c11/dist_tests.log
run() {
  name=$1; shift
  env "$@" timeout -k 10 300 python3 tools/exp/rccl_world1.py -- --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > gpurun_out/r03c11/rccl_$name.txt 2>&1 || { echo "$name FAILED"; tail -5 gpurun_out/r03c11/rccl_$name.txt; return 1; }
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
for line in open(f'gpurun_out/r03c11/rccl_{name}.txt'):
    if line.startswith('{"metric"'):
        d = json.loads(line)
        print(f'{name:28s} {d["ms_per_step"]:7.2f} ms/step  host {d["host_enqueue_ms_per_step"]:6.2f} ms  grad collectives/step {d["config"].get("grad_collectives_per_step")}  layout {d["config"].get("stream_layout_check")}')
    if line.startswith('[rccl_world1]'):
        print('   ', line.strip())
PY
}
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain (no process group)', d['ms_per_step'])"
run r3_default A=1
run r2_behaviour S4F_BUCKET_MIN_ELEMS=0 S4F_AUX_LOCKSTEP=0 S4F_DECODE_LOCKSTEP=0
run r3_no_lockstep S4F_AUX_LOCKSTEP=0 S4F_DECODE_LOCKSTEP=0
run r3_default_b A=1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain (no process group)', d['ms_per_step'])"
