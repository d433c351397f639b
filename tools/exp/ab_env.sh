#!/bin/bash
# same-box A/B of one environment switch: tools/exp/ab_env.sh VAR=a VAR=b [rounds] [bench args...]   (interleaved runs of bench.py)
cd $GRAFT_REPO_ROOT
A=$1; B=$2; rounds=${3:-3}; shift 3
one() { echo -n "$1 "; env $1 timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode "${@:2}" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_windows'), 'loss', d['losses']['loss'])"; }
for i in $(seq 1 $rounds); do one $A "$@"; one $B "$@"; done
