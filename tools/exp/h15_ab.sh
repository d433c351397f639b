#!/bin/bash
# round 5: the step with subsets of its N = 768 launches on the 256 x 192 ping-pong tile (tuned-table overlays a / b) against the shipped table
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for cfg in "A=1" "S4F_TUNE_CACHE=$GRAFT_REPO_ROOT/tools/exp/tuned_h15_a.json" "S4F_TUNE_CACHE=$GRAFT_REPO_ROOT/tools/exp/tuned_h15_b.json"; do
  echo -n "[${cfg##*/}] "; env $cfg timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_windows'])"
done; done
