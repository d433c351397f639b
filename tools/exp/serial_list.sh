#!/bin/bash
# one-stream step under the kernel trace: per-kernel report + launch list   (tools/exp/serial_list.sh <out dir under gpurun_out>)
out=$GRAFT_REPO_ROOT/$1; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
S4F_SIDE_STREAM=0 S4F_HEAD_STREAMS=0 S4F_EAGER_SGD=0 rocprofv3 --kernel-trace -d "$out/serial" -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile > "$out/serial.log" 2>&1 || { tail -5 "$out/serial.log"; exit 1; }
db=$(find "$out/serial" -name '*.db' | head -1)
python3 $R/tools/trace_report.py "$db" > "$out/serial_step_report.txt"; python3 $R/tools/launch_list.py "$db" > "$out/serial_launch_list.txt"
find "$out" -name '*.db' -delete; rm -rf "$out/serial"
