#!/bin/bash
# round 5: the layer's grouped weight gradient on a tile variant with a smaller register / LDS footprint (hint 2: 256 x 128, 8 waves,
# ~90 VGPRs, ~100 KB LDS) so that an HBM-bound chain kernel (ln_bwd) can share its CUs - step time + ln_bwd's in-step average
cd $GRAFT_REPO_ROOT
for cfg in "" "S4F_WG_HINT=2 S4F_WG_SPLITK=2" "S4F_WG_HINT=2 S4F_WG_SPLITK=3" "S4F_WG_HINT=2 S4F_WG_SPLITK=4" "" "S4F_WG_HINT=2 S4F_WG_SPLITK=2"; do
  echo -n "[$cfg] "; env $cfg timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_windows'])"
done
