#!/bin/bash
# hunt for the round-1 2-rank hang: bench.py (DeiT-B, 2 ranks on one GPU over gloo), stacks of a stalled rank to files
# usage: tools/exp/hang_hunt.sh <out dir> <runs> [ENV=VAL ...]
set -u
out=$1; runs=$2; shift 2
mkdir -p "$out"
for i in $(seq 1 "$runs"); do
  d="$out/run$i"; mkdir -p "$d"
  echo "== run $i $(date +%T) $*"
  env "$@" S4F_DIST_BACKEND=gloo S4F_BENCH_WATCHDOG=100 S4F_WATCHDOG_DIR="$d" S4F_DIST_TIMEOUT_S=120 HSA_ENABLE_IPC_MODE_LEGACY=0 \
    timeout -k 10 170 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700+i)) \
    bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > "$d/out.log" 2> "$d/err.log"
  rc=$?
  echo "rc=$rc $(date +%T)"; grep -h '"metric"' "$d/out.log" | cut -c1-200
  if [ $rc -ne 0 ]; then tail -5 "$d/err.log"; exit $rc; fi
done
