#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r03c4; mkdir -p "$out"; cd $GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.json
for r in fp32 auto; do
  S4F_RESID=$r timeout -k 10 900 python3 -m pytest tests/test_step_gpu.py -q -k "step_vs_golden" > "$out/step_$r.log" 2>&1; echo "step $r rc=$?"; tail -3 "$out/step_$r.log"
  S4F_RESID=$r timeout -k 10 1100 python3 -m pytest tests/test_fullsize_gpu.py -q -k "reference_golden" > "$out/full_$r.log" 2>&1; echo "full $r rc=$?"; tail -3 "$out/full_$r.log"
done
cp gpurun_out/parity_report.json "$out/parity_report.json"
