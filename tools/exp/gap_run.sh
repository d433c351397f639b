#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/$1; mkdir -p $out
rm -rf /tmp/gap_tl
rocprofv3 --kernel-trace -d /tmp/gap_tl -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > $out/gap_bench.log 2>&1 || { tail -5 $out/gap_bench.log; exit 1; }
db=$(find /tmp/gap_tl -name '*.db' | head -1)
python3 $R/tools/exp/gap_report.py "$db" 2 > $out/gaps.txt
python3 $R/tools/timeline_report.py "$db" 2 > $out/timeline.txt
tail -1 $out/gap_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('traced ms/step', d['ms_per_step'])"
