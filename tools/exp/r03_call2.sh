#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c2; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --timeline "$out/timeline.json" > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host_enqueue_ms_per_step'], d['host_enqueue_idle_queue_ms'])"
python3 $R/tools/event_timeline.py "$out/timeline.json" > "$out/event_timeline.txt"
head -5 "$out/event_timeline.txt"; echo done
