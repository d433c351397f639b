#!/bin/bash
# round-3 first measurement: default bench + one serialized step, per-launch listing
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c1; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench default $(date +%T)"
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" > "$out/bench_default.json"
cp $R/gpurun_out/bench_kernels_semi_bf16.json "$out/bench_kernel_events.json"
echo "== serial step $(date +%T)"
S4F_SIDE_STREAM=0 S4F_HEAD_STREAMS=0 S4F_EAGER_SGD=0 rocprofv3 --kernel-trace -d "$out/serial" -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile > "$out/serial.log" 2>&1 || { tail -5 "$out/serial.log"; exit 1; }
db=$(find "$out/serial" -name '*.db' | head -1)
python3 $R/tools/trace_report.py "$db" > "$out/serial_step_report.txt"
python3 $R/tools/trace_shapes.py "$db" > "$out/serial_gemm_shapes.txt"
python3 $R/tools/launch_list.py "$db" > "$out/serial_launch_list.txt"
find "$out" -name '*.db' -delete; rm -rf "$out/serial"
tail -1 "$out/bench.log"; echo done
