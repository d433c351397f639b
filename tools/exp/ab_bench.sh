#!/bin/bash
# same-box A/B of two trees: ab_old/ (a built copy of an earlier commit) against the working tree, alternating runs
# usage (GPU box): bash tools/exp/ab_bench.sh [rounds] [bench args...]
rounds=${1:-3}; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $rounds); do
  for side in old new; do
    if [ $side = old ]; then dir=ab_old; else dir=.; fi
    (cd $dir && timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline "$@" 2>/dev/null | grep '^{"metric"' > /tmp/ab_line.json) || { echo "$side failed"; exit 1; }
    python - $side $r <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
print(f'{sys.argv[1]:4s} round {sys.argv[2]}: {d["ms_per_step"]:7.3f} ms/step  loss {d["losses"]["loss"]:.4f}', flush=True)
PY
  done
done
