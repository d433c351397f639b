#!/bin/bash
# round 6: resident-workgroup cap of the weight-gradient launches (S4F_WG_CAP: the layers' grouped launch, S4F_CONVWG_CAP: the
# head convs' launches) and the fork point of the grouped launch (S4F_WG_LATE), interleaved with the default on one box
cd $GRAFT_REPO_ROOT
one() { echo -n "[$1] "; env $1 timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_windows'), 'loss', d['losses']['loss'])"; }
for cfg in "$@"; do one "$cfg"; done
