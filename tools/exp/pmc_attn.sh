#!/bin/bash
# SQ counters of the attention kernels (rocprofv3 --pmc in its own run, kernel-trace only): tools/exp/pmc_attn.sh <out dir> [attn|attnb|attnf]
out=$1; what=${2:-attn}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d "$out/p1" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/one_kernel.py" $what 0 5 > "$out/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$out/p2" -f csv -- python3 "$GRAFT_REPO_ROOT/tools/one_kernel.py" $what 0 5 > "$out/p2.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ('p1', 'p2'):
    for f in glob.glob(f'{out}/{p}/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'attn' in r['Kernel_Name'] or 'fb::' in r['Kernel_Name'] or 'ff::' in r['Kernel_Name'] or '2fb' in r['Kernel_Name'] or '2ff' in r['Kernel_Name']:
                agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, d in agg.items():
            print(p, k)
            for c, v in sorted(d.items()):
                print(f'    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})')
PY
