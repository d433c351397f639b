#!/bin/bash
# interleaved bench.py A/B over environment settings for another workload: tools/exp/r06_env_ab_workload.sh <workload> "ENV=a" "ENV=b" ...
cd $GRAFT_REPO_ROOT
w=$1; shift
one() { echo -n "[$w $1] "; env $1 timeout -k 10 250 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_windows'))"; }
for cfg in "$@"; do one "$cfg"; done
