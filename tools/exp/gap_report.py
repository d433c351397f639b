"""Where is NOTHING running?  From a rocprofv3 --kernel-trace rocpd database: the idle gaps (no queue busy) of one step, longest
first, each with the kernel that ended before it and the one that started after it.   python tools/exp/gap_report.py <results.db> [steps_back] [min_us]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
rows = list(c.execute("select name, start, end, queue_id from kernels order by start"))
em = [r[1] for r in rows if 'ema_kernel' in r[0]]
if len(em) < 4:
    em = sorted(r[1] for r in rows if 'im2col16' in r[0])[0::2]
t0, t1 = em[-k - 1], em[-k]
R = [r for r in rows if t0 <= r[1] < t1]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', n)
    return (m.group(1) if m else n)[:44]


gaps = []
cur_end, last = R[0][2], R[0]
for r in R[1:]:
    if r[1] > cur_end:
        gaps.append((r[1] - cur_end, cur_end, last, r))
    if r[2] > cur_end:
        cur_end, last = r[2], r
tot = sum(g[0] for g in gaps)
print(f'step {(t1 - t0) / 1e6:.2f} ms, {len(R)} kernels, idle (no queue busy) {tot / 1e6:.3f} ms in {len(gaps)} gaps; '
      f'{sum(g[0] for g in gaps if g[0] >= min_us * 1e3) / 1e6:.3f} ms in gaps >= {min_us} us')
hist = {}
for g in gaps:
    b = 1 if g[0] < 2e3 else 2 if g[0] < 5e3 else 5 if g[0] < 10e3 else 10 if g[0] < 20e3 else 20 if g[0] < 50e3 else 50
    hist[b] = hist.get(b, 0) + g[0]
print('idle ms by gap length (us bucket start):', {b: round(v / 1e6, 3) for b, v in sorted(hist.items())})
for g in sorted(gaps, key=lambda g: -g[0])[:40]:
    print(f'{g[0] / 1e3:8.1f} us at {(g[1] - t0) / 1e6:7.3f} ms | after {short(g[2][0])} (q{g[2][3]}) | before {short(g[3][0])} (q{g[3][3]})')
