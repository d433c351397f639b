#!/bin/bash
# the two PMC passes of tools/profile_round.sh alone + the per-kernel traffic table
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/traffic; mkdir -p "$out"; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$out/pmc_fetch" -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-kernel-profile > "$out/pmc_fetch.log" 2>&1 || { tail -5 "$out/pmc_fetch.log"; exit 1; }
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$out/pmc_write" -- python3 $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-kernel-profile > "$out/pmc_write.log" 2>&1 || { tail -5 "$out/pmc_write.log"; exit 1; }
ff=$(find "$out/pmc_fetch" -name '*counter_collection.csv' | head -1); fw=$(find "$out/pmc_write" -name '*counter_collection.csv' | head -1)
head -1 "$ff" > "$out/header.txt"
python3 $R/tools/pmc_traffic.py "$ff" "$fw" "$out/r03_gemm_hbm_traffic.json" "$out/r03_hbm_traffic_by_kernel.txt" > /dev/null
rm -rf "$out/pmc_fetch" "$out/pmc_write"
head -50 "$out/r03_hbm_traffic_by_kernel.txt"
