#!/bin/bash
# gpurun with a retry on exit code 3 only (no box / slot free: nothing ran, nothing charged).  usage: tools/gpu_retry.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
