#!/bin/bash
# kernel-trace timeline of the default step under two settings of one environment switch: tools/exp/timeline_ab.sh VAR=a VAR=b <out dir>
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/$3; mkdir -p $out
for kv in $1 $2; do
  export $kv
  rm -rf /tmp/tl_$kv
  rocprofv3 --kernel-trace -d /tmp/tl_$kv -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > $out/tl_$kv.log 2>&1 || { tail -5 $out/tl_$kv.log; exit 1; }
  db=$(find /tmp/tl_$kv -name '*.db' | head -1)
  python3 $R/tools/timeline_report.py "$db" > $out/timeline_$kv.txt
  tail -1 $out/tl_$kv.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', d['ms_per_step'])"
done
