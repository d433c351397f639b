#!/bin/bash
cd $GRAFT_REPO_ROOT
run() {
  echo -n "$* : "
  env "$@" timeout -k 10 300 python3 bench.py --workload semi768 --steps 15 --warmup 4 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_windows'], d['host_enqueue_idle_queue_ms'])"
}
run A=1
run S4F_FUSED_LAUNCH=0
run S4F_FUSED_ZERO_GRAD=0
run S4F_FUSED_LAUNCH=0 S4F_FUSED_ZERO_GRAD=0
run A=1
