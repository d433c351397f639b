#!/bin/bash
# SQ counters of the fc2-shaped GEMM: this package's 8-wave ping-pong kernel (256 x 256 and 256 x 192 tiles) against hipBLASLt's 4-wave kernel
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_fc2
cd /tmp && export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d "$out/p1" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/exp/r06_pmc_fc2.py" > "$out/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$out/p2" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/exp/r06_pmc_fc2.py" > "$out/p2.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ('p1', 'p2'):
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'gemm5' in n or 'Cijk' in n:
                agg[(n[:70], r.get('Grid_Size', r.get('Grid_Size_X', '')))][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            print(p, k)
            for c, v in sorted(d.items()):
                print(f'    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})')
PY
rm -rf "$out/p1" "$out/p2"
