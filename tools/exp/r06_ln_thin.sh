#!/bin/bash
# round 6: the thin LayerNorm backward (<= 64 registers, co-resident with the 8-wave GEMM workgroups) against the two-rows-per-wave
# kernel: alone / beside the weight-gradient and input-gradient GEMMs (corun_probe), then in the step (interleaved)
cd $GRAFT_REPO_ROOT
for t in 0 1024; do echo "== S4F_LN_BWD_THIN=$t"; S4F_LN_BWD_THIN=$t CORUN_ONLY=ln_bwd timeout -k 10 200 python3 tools/exp/corun_probe.py 2>/dev/null | grep -v "^library"; done
one() { echo -n "[$1] "; env $1 timeout -k 10 250 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --no-parity-mode 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('ms_per_step_windows'), 'loss', d['losses']['loss'])"; }
for cfg in "S4F_LN_BWD_THIN=0" "S4F_LN_BWD_THIN=1024" "S4F_LN_BWD_THIN=512" "S4F_LN_BWD_THIN=2048" "S4F_LN_BWD_THIN=0" "S4F_LN_BWD_THIN=1024" "S4F_LN_BWD_THIN=768" "S4F_LN_BWD_THIN=0" "S4F_LN_BWD_THIN=1024"; do one "$cfg"; done
