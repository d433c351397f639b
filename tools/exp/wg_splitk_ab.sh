#!/bin/bash
# round 5: k-ranges of the layer's grouped weight gradient (108 tiles x k-ranges blocks of the 8-wave k-major kernel): shorter
# workgroups free their CUs sooner for the chain's kernels (ln_bwd) - step time per setting, interleaved with the default (2)
cd $GRAFT_REPO_ROOT
for cfg in "" "S4F_WG_SPLITK=1" "S4F_WG_SPLITK=3" "S4F_WG_SPLITK=4" "" "S4F_WG_SPLITK=3" "S4F_WG_SPLITK=6"; do
  echo -n "[$cfg] "; env $cfg timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_windows'])"
done
