#!/bin/bash
# SQ counters of upce_bwd (rocprofv3 --pmc in its own runs, kernel-trace only): tools/exp/pmc_upce.sh <out dir>
out=$1
cd /tmp && export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES -d "$out/p1" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/exp/upce_probe.py" > "$out/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD -d "$out/p2" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/exp/upce_probe.py" > "$out/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d "$out/p3" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/exp/upce_probe.py" > "$out/p3.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'upce_bwd' in r['Kernel_Name']:
                agg[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            print(p, k)
            for c, v in sorted(d.items()):
                print(f'    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})')
PY
