#!/bin/bash
# same-box A/B of two builds of the library: tools/exp/ab_lib.sh <other .so> [bench args]
cd $GRAFT_REPO_ROOT
other=$1; shift
one() { echo -n "$1 "; env $2 timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode "${@:3}" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_windows'])"; }
for i in 1 2 3; do one shipped A=1 "$@"; one other S4F_LIB=$GRAFT_REPO_ROOT/$other "$@"; done
