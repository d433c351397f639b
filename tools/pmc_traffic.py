"""HBM traffic per GEMM launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py, reduced as
MI355X_MICROARCH.md prescribes: separate passes, FETCH_SIZE doubled on gfx950 (128-B requests tallied at 64 B), units KB.
  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import hashlib
import json
import os
import sys


def kernel_src_sha():
    """hash of the GEMM kernel sources the traffic was measured on: bench.py reports `roofline.traffic` from this file only
    while the hash still matches the tree (a profile that predates a kernel change is refused, loudly)"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 's4former_amd', 'csrc')
    h = hashlib.sha256()
    for fn in ('common.h', 'gemm.hip', 'gemm2.hip', 'gemm5.hip', 'gemm6.hip'):
        h.update(open(os.path.join(root, fn), 'rb').read())
    return h.hexdigest()[:16]


def _step_marks(rows):
    """row indices where a step starts: the EMA launch or, with the double-buffered teacher (round 5), the first of the step's two
    patch-embedding gathers"""
    ema = [i for i, r in enumerate(rows) if 'ema_kernel' in r['Kernel_Name']]
    if len(ema) < 3:
        ema = [i for i, r in enumerate(rows) if 'im2col16' in r['Kernel_Name']][0::2]
    return ema


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    # dispatches are in launch order; the EMA kernel marks the start of a step: keep the last complete step
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ema = _step_marks(rows)
    a, b = ema[-2], ema[-1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[a:b]:
        n = r['Kernel_Name']
        if 'gemm' in n:
            agg['gemm'][0] += 1
            agg['gemm'][1] += float(r['Counter_Value'])
    return agg['gemm']


def by_kernel(path, counter):
    """{(kernel, blocks): [launches, KB]} over the last complete step, every kernel (not only the GEMM family)"""
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    ema = _step_marks(rows)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[ema[-2]:ema[-1]]:
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
        name = name.split('(')[0][:44]
        try:
            blocks = int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))
        except (KeyError, ValueError):
            blocks = 0
        k = (name, blocks)
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    return agg


def table(fetch_csv, write_csv, out_txt):
    f, w = by_kernel(fetch_csv, 'FETCH_SIZE'), by_kernel(write_csv, 'WRITE_SIZE')
    rows = []
    for k in sorted(set(f) | set(w)):
        n = f.get(k, w.get(k))[0]
        fb = 2.0 * f.get(k, [0, 0.0])[1] * 1024 / max(n, 1) / 1e6
        wb = w.get(k, [0, 0.0])[1] * 1024 / max(n, 1) / 1e6
        rows.append((n * (fb + wb), k[0], k[1], n, fb, wb))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    with open(out_txt, 'w') as fh:
        fh.write('HBM-side traffic of one step by kernel and grid (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled for\n'
                 'gfx950, MB per launch; FETCH counts requests that left L2 - reads served by the Infinity Cache included).  Step total '
                 f'{tot / 1e3:.1f} GB.\n\n')
        fh.write(f'{"kernel":46s} {"blocks":>8s} {"launches":>8s} {"fetch MB":>10s} {"write MB":>10s} {"step GB":>8s}\n')
        for t, name, blocks, n, fb, wb in rows[:60]:
            fh.write(f'{name:46s} {blocks:8d} {n:8d} {fb:10.1f} {wb:10.1f} {t / 1e3:8.2f}\n')


def main():
    if len(sys.argv) > 4:
        table(sys.argv[1], sys.argv[2], sys.argv[4])
    fetch_n, fetch_kb = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write_n, write_kb = per_kernel(sys.argv[2], 'WRITE_SIZE')
    assert fetch_n == write_n, (fetch_n, write_n)
    fb = 2.0 * fetch_kb * 1024 / fetch_n
    wb = write_kb * 1024 / write_n
    out = dict(gemm_launches_per_step=fetch_n, FETCH_SIZE_KB_step=fetch_kb, WRITE_SIZE_KB_step=write_kb,
               fetch_bytes_per_launch=fb, write_bytes_per_launch=wb, hbm_bytes_per_launch=fb + wb, kernel_src_sha=kernel_src_sha(),
               note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 --warmup 3 (semi, '
                    'bf16); every GEMM-family dispatch (gemm_kernel / gemm2 / gemm5 / gemm6, grouped) of the last complete step; '
                    'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B), units KB')
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
