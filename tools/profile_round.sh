#!/bin/bash
# The profiles of a round, on the GPU box:  tools/profile_round.sh <out dir under gpurun_out> <tag e.g. r02>
# (rocprofv3 counters in their own passes, kernel-trace/stats only; programs directly after `--`)
set -u
out=$GRAFT_REPO_ROOT/$1; tag=$2
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== HBM traffic of the GEMM family (two pmc passes) $(date +%T)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$out/pmc_fetch" -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > "$out/pmc_fetch.log" 2>&1 || { tail -5 "$out/pmc_fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$out/pmc_write" -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > "$out/pmc_write.log" 2>&1 || { tail -5 "$out/pmc_write.log"; exit 1; }
ff=$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1); fw=$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)
python3 $R/tools/pmc_traffic.py "$ff" "$fw" "$out/${tag}_gemm_hbm_traffic.json" "$out/${tag}_hbm_traffic_by_kernel.txt" > /dev/null
cp "$out/${tag}_gemm_hbm_traffic.json" $R/profiles/     # (the box's copy) so that the bench line below reports roofline.traffic of THESE kernel sources
echo "== kernel stats $(date +%T)"
rocprofv3 --kernel-trace --stats -f csv -d "$out/stats" -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode > "$out/stats.log" 2>&1 || { tail -5 "$out/stats.log"; exit 1; }
f=$(find "$out/stats" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_rocprofv3_kernel_stats_semi_bf16.csv"
echo "== serial step (one stream) $(date +%T)"
S4F_SIDE_STREAM=0 S4F_HEAD_STREAMS=0 S4F_EAGER_SGD=0 rocprofv3 --kernel-trace -d "$out/serial" -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > "$out/serial.log" 2>&1 || { tail -5 "$out/serial.log"; exit 1; }
db=$(find "$out/serial" -name '*.db' | head -1); [ -n "$db" ] && python3 $R/tools/trace_report.py "$db" > "$out/${tag}_serial_step_report.txt" && python3 $R/tools/trace_shapes.py "$db" > "$out/${tag}_serial_gemm_shapes.txt" 2>/dev/null && python3 $R/tools/launch_list.py "$db" > "$out/${tag}_serial_launch_list.txt"
# GEMM-family kernel time per step (traced / serial) for bench.py's roofline.traced / roofline.serial; the box's copy goes to profiles/ at once
(cd $R/tools && python3 gemm_time_profile.py "$out/${tag}_rocprofv3_kernel_stats_semi_bf16.csv" "$out/${tag}_serial_step_report.txt" "$out/${tag}_gemm_kernel_time.json" > /dev/null) && cp "$out/${tag}_gemm_kernel_time.json" $R/profiles/
echo "== bench default $(date +%T)"
python3 $R/bench.py --steps 20 --warmup 5 --timeline "$out/timeline_events.json" > "$out/bench.log" 2>&1 || { tail -5 "$out/bench.log"; exit 1; }
tail -1 "$out/bench.log" > "$out/${tag}_bench_default.json"
cp $R/gpurun_out/bench_kernels_semi_bf16.json "$out/${tag}_bench_kernel_events_semi_bf16.json"
python3 $R/tools/event_timeline.py "$out/timeline_events.json" > "$out/${tag}_event_timeline_untraced.txt" && rm -f "$out/timeline_events.json"
echo "== timeline (default streams) $(date +%T)"
rocprofv3 --kernel-trace -d "$out/timeline" -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-profile --no-parity-mode > "$out/timeline.log" 2>&1 || { tail -5 "$out/timeline.log"; exit 1; }
db=$(find "$out/timeline" -name '*.db' | head -1); [ -n "$db" ] && python3 $R/tools/timeline_report.py "$db" > "$out/${tag}_timeline_default.txt"
echo "== attention $(date +%T)"
python3 $R/tools/attn_probe.py > "$out/${tag}_attention_probe.txt" 2>&1
$R/tools/exp/pmc_attn.sh "$out/pmc_attn_fwd" attn > "$out/${tag}_pmc_attention_fwd.txt" 2>&1
$R/tools/exp/pmc_attn.sh "$out/pmc_attn_bwd" attnf > "$out/${tag}_pmc_attention_bwd.txt" 2>&1     # the one-sweep backward (round 4)
echo "== other workloads $(date +%T)"
for w in sup semi768 ours; do python3 $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-parity-mode 2>/dev/null | tail -1 > "$out/${tag}_bench_${w}.json"; done
# keep the merged-back payload small: databases and raw traces stay on the box
find "$out" -name '*.db' -delete; rm -rf "$out/stats" "$out/serial" "$out/timeline" "$out/pmc_fetch" "$out/pmc_write" "$out"/pmc_attn_*/p1 "$out"/pmc_attn_*/p2
ls -la "$out"; echo done
