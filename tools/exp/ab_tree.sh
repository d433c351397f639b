#!/bin/bash
# same-box A/B of two source trees: ab_base/ (git archive of a commit, built in place) against the working tree
cd $GRAFT_REPO_ROOT
one() { (cd $1 && timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode ${ARGS:-} 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['ms_per_step_windows'])"); }
for i in 1 2 3; do one ab_base; one .; done
